// Image output by file extension, standing in for `image.save(file_name)` of the image crate
// (src/render.rs:273-276), and the PPM reader used for the pre-decoded earth map
// (the reference decodes earthmap.jpg with the image crate, src/texture.rs:98).
#include <zlib.h>

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
  else { rth::g_err = "The image format could not be determined (supported: .ppm, .png)"; return 2; }
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
