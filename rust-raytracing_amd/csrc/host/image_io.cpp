// Image output by file extension, standing in for `image.save(file_name)` of the image crate
// (src/render.rs:273-276), and the PPM reader used for the pre-decoded earth map
// (the reference decodes earthmap.jpg with the image crate, src/texture.rs:98).
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_util.h"
#include "rtamd_host.h"

namespace {

bool ends_with(const std::string& s, const char* suf) {
  size_t n = std::strlen(suf);
  if (s.size() < n) return false;
  for (size_t i = 0; i < n; ++i) {
    char a = s[s.size() - n + i], b = suf[i];
    if (a >= 'A' && a <= 'Z') a = (char)(a - 'A' + 'a');
    if (a != b) return false;
  }
  return true;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void png_chunk(std::vector<uint8_t>& out, const char tag[4], const uint8_t* data, size_t n) {
  put_be32(out, (uint32_t)n);
  size_t start = out.size();
  out.insert(out.end(), tag, tag + 4);
  if (n) out.insert(out.end(), data, data + n);
  uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4));
  put_be32(out, crc);
}

int write_png(const char* path, const uint8_t* rgb, uint32_t w, uint32_t h) {
  std::vector<uint8_t> raw;
  raw.reserve((size_t)h * (3 * (size_t)w + 1));
  for (uint32_t y = 0; y < h; ++y) {
    raw.push_back(0);  // filter type None
    raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return 1;
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, w); put_be32(ihdr, h);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // 8-bit RGB
  png_chunk(out, "IHDR", ihdr.data(), ihdr.size());
  png_chunk(out, "IDAT", z.data(), zlen);
  png_chunk(out, "IEND", nullptr, 0);
  FILE* f = std::fopen(path, "wb");
  if (!f) return 1;
  size_t wr = std::fwrite(out.data(), 1, out.size(), f);
  std::fclose(f);
  return wr == out.size() ? 0 : 1;
}

int write_ppm(const char* path, const uint8_t* rgb, uint32_t w, uint32_t h) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return 1;
  std::fprintf(f, "P6\n%u %u\n255\n", w, h);
  size_t n = (size_t)w * h * 3;
  size_t wr = std::fwrite(rgb, 1, n, f);
  std::fclose(f);
  return wr == n ? 0 : 1;
}

// Baseline JPEG (JFIF, 8-bit, three components without subsampling, the standard's example quantisation and Huffman tables,
// quality 75): what `image.save("x.jpg")` produces through the image crate's JpegEncoder at its default quality.  The crate is
// not part of the reference's tree, so the bytes are not compared with its output; the tests decode the file again.
struct JpegBits {
  std::vector<uint8_t>& out;
  uint32_t acc = 0;
  int n = 0;
  void put(uint32_t code, int len) {
    acc = (acc << len) | (code & ((1u << len) - 1u));
    n += len;
    while (n >= 8) {
      uint8_t b = (uint8_t)(acc >> (n - 8));
      out.push_back(b);
      if (b == 0xFF) out.push_back(0);  // byte stuffing
      n -= 8;
    }
  }
  void flush() { if (n) put(0x7F, 8 - n); }
};

struct JpegHuff { uint16_t code[256]; uint8_t len[256]; };

JpegHuff jpeg_huff(const uint8_t counts[16], const uint8_t* symbols) {
  JpegHuff h{};
  uint32_t code = 0;
  int k = 0;
  for (int l = 1; l <= 16; ++l) {
    for (int i = 0; i < counts[l - 1]; ++i, ++k) { h.code[symbols[k]] = (uint16_t)code++; h.len[symbols[k]] = (uint8_t)l; }
    code <<= 1;
  }
  return h;
}

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                             15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kQLuma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57,
                            69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64,
                            81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kQChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                              99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kDcLumaN[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaN[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcSym[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaN[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaSym[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaN[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaSym[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

void jpeg_marker(std::vector<uint8_t>& o, uint8_t m, size_t payload) {
  o.push_back(0xFF); o.push_back(m);
  o.push_back((uint8_t)((payload + 2) >> 8)); o.push_back((uint8_t)(payload + 2));
}

void jpeg_dht(std::vector<uint8_t>& o, uint8_t cls_id, const uint8_t counts[16], const uint8_t* sym) {
  size_t n = 0;
  for (int i = 0; i < 16; ++i) n += counts[i];
  jpeg_marker(o, 0xC4, 1 + 16 + n);
  o.push_back(cls_id);
  o.insert(o.end(), counts, counts + 16);
  o.insert(o.end(), sym, sym + n);
}

// Separable 8-point DCT-II of one block (orthonormal scaling folded into `basis`), in place.
void jpeg_fdct(float* blk, const float basis[8][8]) {
  float tmp[64];
  for (int y = 0; y < 8; ++y)
    for (int u = 0; u < 8; ++u) {
      float s = 0;
      for (int x = 0; x < 8; ++x) s += blk[y * 8 + x] * basis[u][x];
      tmp[y * 8 + u] = s;
    }
  for (int u = 0; u < 8; ++u)
    for (int v = 0; v < 8; ++v) {
      float s = 0;
      for (int y = 0; y < 8; ++y) s += tmp[y * 8 + u] * basis[v][y];
      blk[v * 8 + u] = s;
    }
}

void jpeg_block(JpegBits& bw, float* blk, const float basis[8][8], const uint8_t* q, int* dc_prev, const JpegHuff& hdc,
                const JpegHuff& hac) {
  jpeg_fdct(blk, basis);
  int zz[64];
  for (int i = 0; i < 64; ++i) {
    float v = blk[kZigzag[i]] / (float)q[kZigzag[i]];
    zz[i] = (int)(v < 0 ? v - 0.5f : v + 0.5f);
  }
  auto magnitude = [](int v, int* bits) {
    int a = v < 0 ? -v : v, n = 0;
    while (a) { ++n; a >>= 1; }
    *bits = v < 0 ? v - 1 : v;
    return n;
  };
  int bits, diff = zz[0] - *dc_prev;
  *dc_prev = zz[0];
  int n = magnitude(diff, &bits);
  bw.put(hdc.code[n], hdc.len[n]);
  if (n) bw.put((uint32_t)bits, n);
  int last = 63;
  while (last > 0 && zz[last] == 0) --last;
  int run = 0;
  for (int i = 1; i <= last; ++i) {
    if (zz[i] == 0) { ++run; continue; }
    while (run >= 16) { bw.put(hac.code[0xF0], hac.len[0xF0]); run -= 16; }
    n = magnitude(zz[i], &bits);
    int sym = (run << 4) | n;
    bw.put(hac.code[sym], hac.len[sym]);
    bw.put((uint32_t)bits, n);
    run = 0;
  }
  if (last < 63) bw.put(hac.code[0], hac.len[0]);  // end of block
}

int write_jpeg(const char* path, const uint8_t* rgb, uint32_t w, uint32_t h) {
  if (w > 65535 || h > 65535) return 1;
  const int quality = 75, scale = 200 - 2 * quality;  // the usual quality-to-scale rule above 50
  uint8_t q[2][64];
  for (int i = 0; i < 64; ++i) {
    int a = (kQLuma[i] * scale + 50) / 100, b = (kQChroma[i] * scale + 50) / 100;
    q[0][i] = (uint8_t)(a < 1 ? 1 : a > 255 ? 255 : a);
    q[1][i] = (uint8_t)(b < 1 ? 1 : b > 255 ? 255 : b);
  }
  std::vector<uint8_t> o = {0xFF, 0xD8};
  jpeg_marker(o, 0xE0, 14);
  const uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 2, 0, 0, 1, 0, 1, 0, 0};
  o.insert(o.end(), jfif, jfif + 14);
  for (int t = 0; t < 2; ++t) {
    jpeg_marker(o, 0xDB, 65);
    o.push_back((uint8_t)t);
    for (int i = 0; i < 64; ++i) o.push_back(q[t][kZigzag[i]]);
  }
  jpeg_marker(o, 0xC0, 15);
  const uint8_t sof[15] = {8, (uint8_t)(h >> 8), (uint8_t)h, (uint8_t)(w >> 8), (uint8_t)w, 3, 1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1};
  o.insert(o.end(), sof, sof + 15);
  jpeg_dht(o, 0x00, kDcLumaN, kDcSym);
  jpeg_dht(o, 0x10, kAcLumaN, kAcLumaSym);
  jpeg_dht(o, 0x01, kDcChromaN, kDcSym);
  jpeg_dht(o, 0x11, kAcChromaN, kAcChromaSym);
  jpeg_marker(o, 0xDA, 10);
  const uint8_t sos[10] = {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
  o.insert(o.end(), sos, sos + 10);

  float basis[8][8];
  for (int u = 0; u < 8; ++u)
    for (int x = 0; x < 8; ++x)
      basis[u][x] = (u ? 0.5f : 0.35355339f) * std::cos((2 * x + 1) * u * 3.14159265358979f / 16.0f);
  JpegHuff hdc[2] = {jpeg_huff(kDcLumaN, kDcSym), jpeg_huff(kDcChromaN, kDcSym)};
  JpegHuff hac[2] = {jpeg_huff(kAcLumaN, kAcLumaSym), jpeg_huff(kAcChromaN, kAcChromaSym)};
  JpegBits bw{o};
  int dc[3] = {0, 0, 0};
  for (uint32_t by = 0; by < h; by += 8)
    for (uint32_t bx = 0; bx < w; bx += 8) {
      float blk[3][64];
      for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
          uint32_t py = by + y < h ? by + y : h - 1, px = bx + x < w ? bx + x : w - 1;  // edge pixels repeated
          const uint8_t* p = rgb + ((size_t)py * w + px) * 3;
          float r = p[0], g = p[1], b = p[2];
          blk[0][y * 8 + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
          blk[1][y * 8 + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
          blk[2][y * 8 + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
        }
      for (int c = 0; c < 3; ++c) jpeg_block(bw, blk[c], basis, q[c ? 1 : 0], &dc[c], hdc[c ? 1 : 0], hac[c ? 1 : 0]);
    }
  bw.flush();
  o.push_back(0xFF); o.push_back(0xD9);
  FILE* f = std::fopen(path, "wb");
  if (!f) return 1;
  size_t wr = std::fwrite(o.data(), 1, o.size(), f);
  std::fclose(f);
  return wr == o.size() ? 0 : 1;
}

int ppm_token(FILE* f, long* out) {  // next whitespace-separated integer, skipping '#' comments
  int c = std::fgetc(f);
  for (;;) {
    while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = std::fgetc(f);
    if (c != '#') break;
    while (c != '\n' && c != EOF) c = std::fgetc(f);
  }
  if (c < '0' || c > '9') return 1;
  long v = 0;
  while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); c = std::fgetc(f); }
  *out = v;  // the single whitespace after the token has been consumed
  return 0;
}

}  // namespace

extern "C" {

int rth_save_image(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
  if (!path || !rgb8 || !width || !height) { rth::g_err = "rth_save_image: bad argument"; return 1; }
  std::string p(path);
  int rc;
  if (ends_with(p, ".png")) rc = write_png(path, rgb8, width, height);
  else if (ends_with(p, ".ppm")) rc = write_ppm(path, rgb8, width, height);
  else if (ends_with(p, ".jpg") || ends_with(p, ".jpeg")) rc = write_jpeg(path, rgb8, width, height);
  else { rth::g_err = "The image format could not be determined (supported: .ppm, .png, .jpg)"; return 2; }
  if (rc) rth::g_err = std::string("could not write ") + path;
  return rc;
}

int rth_load_ppm(const char* path, uint8_t** rgb8_out, int32_t* width, int32_t* height) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { rth::g_err = std::string("cannot open ") + path; return 1; }
  int rc = 1;
  long w = 0, h = 0, maxv = 0;
  if (std::fgetc(f) == 'P' && std::fgetc(f) == '6' && !ppm_token(f, &w) && !ppm_token(f, &h) && !ppm_token(f, &maxv) && maxv == 255 &&
      w > 0 && h > 0 && w < 65536 && h < 65536) {
    size_t n = (size_t)w * h * 3;
    uint8_t* buf = (uint8_t*)std::malloc(n);
    if (buf && std::fread(buf, 1, n, f) == n) {
      *rgb8_out = buf; *width = (int32_t)w; *height = (int32_t)h;
      rc = 0;
    } else {
      std::free(buf);
    }
  }
  std::fclose(f);
  if (rc) rth::g_err = std::string("not a binary 8-bit PPM: ") + path;
  return rc;
}

void rth_free(void* p) { std::free(p); }

}  // extern "C"
