// Device-side construction of the library's own four-wide tree for large scenes (BVHNode::new, src/bvh.rs:19-76, is the
// reference's builder: a random axis per node and a median split on the host; the closest hit does not depend on the tree, so
// the library is free to build another).  Input: one box and one leaf word per object.  Steps, all on the context's stream:
//   1. 63-bit Morton codes of the box centres (21 bits an axis, one scale for all axes so that cells are cubes),
//   2. a radix sort of (code, leaf) pairs (rocPRIM's device-wide sort through hipCUB),
//   3. the four-wide tree top-down over the sorted order, one launch pair per level: a node owns a run of leaves and cuts it into
//      up to four parts.  Runs of up to 128 leaves are cut by the surface-area heuristic over the cuts the sorted order offers
//      (in two, then twice more the part of largest area: the host builder's rule, accel_build.h); longer runs near their
//      quartiles - within 1/9 (runs up to 4096) or 1/32 of the run's length on either side - at the place where the Morton
//      codes change in their most significant bit, i.e. on the largest cell boundary the window holds; runs of four leaves or
//      fewer end the descent.  Parts that hold two leaves or more become nodes, numbered by a prefix sum over the level, so the
//      nodes come out level by level in the order of their parents and the tree is the same on every run,
//   4. boxes bottom-up, one launch per level, and with them the stack the deepest walk can need.  The windows bound the
//      imbalance, so the tree of a million leaves is twelve levels deep and the walk's stack (three entries a level) fits the
//      kernels' 36; should a tree come out deeper, it is cut again with the narrow windows only, then at the exact quartiles.
// A plain radix tree over the codes (one inner node per differing bit) is the usual form of this builder; its height follows
// the codes - forty levels on the million-sphere field, neighbours a tenth of a unit apart - and with it the walk's worst-case
// stack, which is why the cuts here are tied to the quartiles instead.
// Forty-odd small launches and one 4-byte read-back per level; a million spheres take 6 ms.
#include <hipcub/hipcub.hpp>

#include "context.h"

namespace rti {
namespace {

constexpr uint32_t kBuildBlock = 256;

struct F6 { float lo[3], hi[3]; };

__device__ inline uint64_t spread21(uint32_t v) {  // bit k of v -> bit 3k
  uint64_t x = v & 0x1FFFFFull;
  x = (x | (x << 32)) & 0x1F00000000FFFFull;
  x = (x | (x << 16)) & 0x1F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

__global__ void k_morton(const F6* boxes, uint32_t n, float ox, float oy, float oz, float scale, uint64_t* keys, uint32_t* vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const F6 b = boxes[i];
  auto cell = [&](float lo, float hi, float o) {
    const float c = (0.5f * (lo + hi) - o) * scale;
    return (uint32_t)fminf(fmaxf(c, 0.0f), 2097151.0f);
  };
  keys[i] = (spread21(cell(b.lo[0], b.hi[0], ox)) << 2) | (spread21(cell(b.lo[1], b.hi[1], oy)) << 1) | spread21(cell(b.lo[2], b.hi[2], oz));
  vals[i] = i;
}

__global__ void k_gather_boxes(const F6* boxes, const uint32_t* sorted_leaf, uint32_t n, F6* sorted_boxes) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sorted_boxes[i] = boxes[sorted_leaf[i]];
}

__device__ inline float half_area(const F6& b) {
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  return dx * dy + dy * dz + dz * dx;
}
__device__ inline void grow(F6& a, const F6& b) {
  for (int k = 0; k < 3; ++k) { a.lo[k] = fminf(a.lo[k], b.lo[k]); a.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
}

// Where to cut the run [l, r) of sorted leaves in two: the position that minimises area(left) * |left| + area(right) * |right|
// (the surface-area heuristic over the cuts the sorted order offers), each side keeping at least a sixth of the run.  `suffix` is
// scratch for the areas of the run's tails, indexed like the leaves (runs of one level are disjoint).
__device__ inline uint32_t sah_cut(const F6* sorted_boxes, uint32_t l, uint32_t r, float* suffix, float& area_l, float& area_r) {
  const uint32_t cnt = r - l;
  const float inf = __int_as_float(0x7F800000);
  F6 acc;
  for (int k = 0; k < 3; ++k) { acc.lo[k] = inf; acc.hi[k] = -inf; }
  for (uint32_t i = r; i-- > l + 1u;) { grow(acc, sorted_boxes[i]); suffix[i] = half_area(acc); }
  for (int k = 0; k < 3; ++k) { acc.lo[k] = inf; acc.hi[k] = -inf; }
  const uint32_t margin = cnt / 6u > 1u ? cnt / 6u : 1u;
  float best = inf;
  uint32_t cut = l + cnt / 2u;
  area_l = area_r = 0.0f;
  for (uint32_t i = l; i + 1u < r; ++i) {
    grow(acc, sorted_boxes[i]);
    const uint32_t nl = i - l + 1u, nr = cnt - nl;
    if (nl < margin || nr < margin) continue;
    const float al = half_area(acc), ar = suffix[i + 1u];
    const float cost = al * (float)nl + ar * (float)nr;
    if (cost < best || best == inf) { best = cost; cut = i + 1u; area_l = al; area_r = ar; }
  }
  return cut;
}

// One level of the tree, first half: where the wide nodes [begin, end) cut their runs; node k owns the sorted leaves
// [run_lo[k], run_hi[k]) and writes five boundaries (up to four parts) and the number of parts that hold two leaves or more.  `mode` 0: runs of up to `sah_run` leaves are cut by the surface-area heuristic (in two, then
// twice more the part of largest area), longer ones near their quartiles on the largest cell boundary within cnt/9 (runs up to 4096) or cnt/32;
// 1: quartile windows of cnt/32 only; 2: the exact quartiles.
__global__ void k_split_level(uint32_t begin, uint32_t end, const uint64_t* keys, const uint32_t* sorted_leaf, const uint32_t* leaf_word, const F6* sorted_boxes,
                              float* suffix, int mode, uint32_t sah_run, const uint32_t* run_lo, const uint32_t* run_hi, uint32_t* cuts, uint32_t* n_inner) {
  const uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= end) return;
  const uint32_t lo = run_lo[k], hi = run_hi[k], cnt = hi - lo;
  uint32_t cut[5] = {lo, lo, lo, lo, lo};
  uint32_t m;
  if (cnt <= 4u) {
    m = cnt;
    for (uint32_t c = 0; c <= m; ++c) cut[c] = lo + c;
  } else if (mode == 0 && cnt <= sah_run) {
    // the host builder's rule (accel_build.h) on the cuts the sorted order offers: in two, then twice more the part of largest
    // surface area that still holds two leaves or more
    float area[4];
    m = 2u;
    cut[0] = lo; cut[2] = hi;
    cut[1] = sah_cut(sorted_boxes, lo, hi, suffix, area[0], area[1]);
    while (m < 4u) {
      int pick = -1;
      float best = -1.0f;
      for (uint32_t j = 0; j < m; ++j)
        if (cut[j + 1] - cut[j] >= 2u && area[j] > best) { best = area[j]; pick = (int)j; }
      if (pick < 0) break;
      float al, ar;
      const uint32_t at = sah_cut(sorted_boxes, cut[pick], cut[pick + 1], suffix, al, ar);
      for (int j = (int)m; j > pick; --j) { cut[j + 1] = cut[j]; if (j < (int)m) area[j + 1] = area[j]; }
      cut[pick + 1] = at;
      area[pick] = al; area[pick + 1] = ar;
      ++m;
    }
  } else {
    m = 4u;
    cut[0] = lo; cut[4] = hi;
    const uint32_t w = (mode == 2 || cnt < 16u) ? 0u : (mode == 0 && cnt <= 4096u ? cnt / 9u : (cnt < 64u ? 0u : cnt / 32u));
    for (uint32_t i = 1; i < 4u; ++i) {
      const uint32_t target = lo + (uint32_t)(((uint64_t)cnt * i) / 4u);
      uint32_t s = target;
      if (w > 0u) {
        const uint32_t a = target - w, b = target + w;  // the cut goes in front of one of the leaves a + 1 .. b
        const uint64_t diff = keys[a] ^ keys[b];
        if (diff != 0ull) {
          const uint64_t bit = 1ull << (63 - __clzll((long long)diff));  // the codes of the window agree above this bit, and it goes 0 -> 1 once
          uint32_t l = a + 1u, r = b;
          while (l < r) {
            const uint32_t mid = (l + r) >> 1;
            if (keys[mid] & bit) r = mid; else l = mid + 1u;
          }
          s = l;
        }
      }
      cut[i] = s;
    }
  }
  uint32_t inner = 0;
  for (uint32_t c = 0; c < m; ++c) inner += cut[c + 1] - cut[c] >= 2u ? 1u : 0u;
  for (uint32_t c = m + 1u; c < 5u; ++c) cut[c] = cut[m];  // unused parts are empty
  for (uint32_t c = 0; c < 5u; ++c) cuts[5u * (k - begin) + c] = cut[c];
  n_inner[k - begin] = inner;
}

// Second half of a level: `first_child[k - begin]` (the exclusive prefix sum of n_inner) numbers the children that are nodes, so
// the nodes of the next level stand in the order of their parents - neighbours in space are neighbours in memory, and the tree
// is the same on every run.
__global__ void k_emit_level(uint32_t begin, uint32_t end, const uint32_t* cuts, const uint32_t* n_inner, const uint32_t* first_child, const uint32_t* sorted_leaf,
                             const uint32_t* leaf_word, uint32_t* run_lo, uint32_t* run_hi, uint32_t* n_wide, WideNode* wide) {
  const uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= end) return;
  uint32_t next = end + first_child[k - begin];
  if (k == end - 1u) *n_wide = next + n_inner[k - begin];
  WideNode nd;
  const float inf = __int_as_float(0x7F800000);
  for (uint32_t c = 0; c < 4u; ++c) {
    nd.pad[c] = 0u;
    for (int a = 0; a < 3; ++a) { nd.lo[a][c] = inf; nd.hi[a][c] = inf; }
    const uint32_t l = cuts[5u * (k - begin) + c], r = cuts[5u * (k - begin) + c + 1u];
    if (r == l) { nd.child[c] = kWideNone; continue; }
    if (r - l == 1u) {
      nd.child[c] = leaf_word[sorted_leaf[l]];
      nd.pad[c] = l;
    } else {
      run_lo[next] = l; run_hi[next] = r;
      nd.child[c] = next++;
    }
  }
  wide[k] = nd;
}

// Boxes of the nodes [begin, end), whose children's own boxes are in place (deeper levels first), and the stack need below each.
__global__ void k_boxes_level(uint32_t begin, uint32_t end, const F6* sorted_boxes, WideNode* wide, F6* node_box, uint32_t* need) {
  const uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= end) return;
  F6 all;
  const float inf = __int_as_float(0x7F800000);
  for (int a = 0; a < 3; ++a) { all.lo[a] = inf; all.hi[a] = -inf; }
  uint32_t m = 0, below = 0;
  for (int c = 0; c < 4; ++c) {
    const uint32_t ch = wide[k].child[c];
    if (ch == kWideNone) continue;
    ++m;
    F6 b;
    if (ch & LEAF_FLAG) {
      b = sorted_boxes[wide[k].pad[c]];
      wide[k].pad[c] = 0u;
    } else {
      b = node_box[ch];
      below = below > need[ch] ? below : need[ch];
    }
    for (int a = 0; a < 3; ++a) {
      wide[k].lo[a][c] = b.lo[a]; wide[k].hi[a][c] = b.hi[a];
      all.lo[a] = fminf(all.lo[a], b.lo[a]); all.hi[a] = fmaxf(all.hi[a], b.hi[a]);
    }
  }
  node_box[k] = all;
  need[k] = (m > 0u ? m - 1u : 0u) + below;  // a node with m children leaves m - 1 of them stacked while the walk is below it
}

}  // namespace

int gpu_sort_pairs(rt_context* c, uint64_t* keys, uint32_t* vals, size_t n) {
  if (n == 0) return RT_OK;
  if (n > 0x7FFFFFFFull) return fail(RT_E_INVALID, "gpu_sort_pairs: too many pairs");
  uint64_t *dk = nullptr, *dk2 = nullptr;
  uint32_t *dv = nullptr, *dv2 = nullptr;
  void* tmp = nullptr;
  auto done = [&](int rc) {
    for (void* p : {(void*)dk, (void*)dk2, (void*)dv, (void*)dv2, tmp}) if (p) (void)hipFree(p);
    return rc;
  };
  size_t tb = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dk2, dv, dv2, (int)n, 0, 64, c->stream) != hipSuccess) return fail(RT_E_HIP, "gpu_sort_pairs: sort set-up failed");
  if (hipMalloc((void**)&dk, 8 * n) != hipSuccess || hipMalloc((void**)&dk2, 8 * n) != hipSuccess || hipMalloc((void**)&dv, 4 * n) != hipSuccess ||
      hipMalloc((void**)&dv2, 4 * n) != hipSuccess || hipMalloc(&tmp, tb) != hipSuccess)
    return done(fail(RT_E_HIP, "gpu_sort_pairs: out of device memory"));
  bool ok = hipMemcpyAsync(dk, keys, 8 * n, hipMemcpyHostToDevice, c->stream) == hipSuccess && hipMemcpyAsync(dv, vals, 4 * n, hipMemcpyHostToDevice, c->stream) == hipSuccess;
  ok = ok && hipcub::DeviceRadixSort::SortPairs(tmp, tb, dk, dk2, dv, dv2, (int)n, 0, 64, c->stream) == hipSuccess;
  ok = ok && hipMemcpyAsync(keys, dk2, 8 * n, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipMemcpyAsync(vals, dv2, 4 * n, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
  ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
  return done(ok ? RT_OK : fail(RT_E_HIP, "gpu_sort_pairs: a HIP call failed"));
}

// boxes: n records of six floats (lo xyz, hi xyz), already rounded outward; words: the leaf word of each; cen_lo / cen_hi: bounds of
// the box centres.  On success *wide_out is a device array of *n_wide_out nodes (hipFree by the caller), root first, level by level.
int gpu_build_wide(rt_context* c, const float* boxes, const uint32_t* words, size_t n_leaves, const double cen_lo[3], const double cen_hi[3],
                   WideNode** wide_out, uint32_t* n_wide_out, int* stack_need_out) {
  *wide_out = nullptr; *n_wide_out = 0; *stack_need_out = 0;
  if (n_leaves < 2 || n_leaves > 0x3FFFFFFFull) return fail(RT_E_INVALID, "gpu_build_wide: between 2 and 2^30 leaves");
  const uint32_t n = (uint32_t)n_leaves;
  hipStream_t st = c->stream;
  struct Pool {  // everything but the result is scratch of this call
    std::vector<void*> p;
    ~Pool() { for (void* q : p) (void)hipFree(q); }
    int get(void** out, size_t bytes) {
      if (hipMalloc(out, bytes) != hipSuccess) return fail(RT_E_HIP, "gpu_build_wide: out of device memory");
      p.push_back(*out);
      return RT_OK;
    }
  } pool;
  F6 *d_boxes, *d_node_box, *d_sorted_boxes; float* d_suffix; uint32_t *d_words, *d_cuts, *d_n_inner, *d_first_child; uint64_t *d_keys, *d_keys_sorted; uint32_t *d_vals, *d_sorted_leaf, *d_run_lo, *d_run_hi, *d_n_wide, *d_need;
#define GET(ptr, bytes) if (int rc_ = pool.get((void**)&(ptr), (bytes))) return rc_
  GET(d_boxes, sizeof(F6) * n_leaves); GET(d_words, 4 * n_leaves);
  GET(d_keys, 8 * n_leaves); GET(d_keys_sorted, 8 * n_leaves); GET(d_vals, 4 * n_leaves); GET(d_sorted_leaf, 4 * n_leaves);
  GET(d_sorted_boxes, sizeof(F6) * n_leaves); GET(d_suffix, 4 * n_leaves);
  GET(d_cuts, 20 * n_leaves); GET(d_n_inner, 4 * n_leaves); GET(d_first_child, 4 * n_leaves);
  GET(d_run_lo, 4 * n_leaves); GET(d_run_hi, 4 * n_leaves); GET(d_n_wide, 4); GET(d_need, 4 * n_leaves); GET(d_node_box, sizeof(F6) * n_leaves);
  WideNode* d_wide = nullptr;  // a node owns at least two leaves, so there are fewer nodes than leaves
  if (hipMalloc((void**)&d_wide, sizeof(WideNode) * n_leaves) != hipSuccess) return fail(RT_E_HIP, "gpu_build_wide: out of device memory");
  auto bail = [&](int rc) { (void)hipFree(d_wide); return rc; };
#define TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(RT_E_HIP, std::string("gpu_build_wide: ") + hipGetErrorString(e_))); } while (0)
  TRY(hipMemcpyAsync(d_boxes, boxes, sizeof(F6) * n_leaves, hipMemcpyHostToDevice, st));
  TRY(hipMemcpyAsync(d_words, words, 4 * n_leaves, hipMemcpyHostToDevice, st));
  double ext = 0.0;
  for (int k = 0; k < 3; ++k) ext = std::max(ext, cen_hi[k] - cen_lo[k]);
  const float scale = ext > 0.0 ? (float)(2097152.0 / ext) : 0.0f;
  const uint32_t grid_n = (n + kBuildBlock - 1) / kBuildBlock;
  hipLaunchKernelGGL(k_morton, dim3(grid_n), dim3(kBuildBlock), 0, st, d_boxes, n, (float)cen_lo[0], (float)cen_lo[1], (float)cen_lo[2], scale, d_keys, d_vals);
  size_t temp_bytes = 0;
  TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys, d_keys_sorted, d_vals, d_sorted_leaf, (int)n, 0, 63, st));
  size_t scan_bytes = 0;
  TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_n_inner, d_first_child, (int)n, st));
  temp_bytes = std::max(temp_bytes, scan_bytes);
  void* d_temp;
  GET(d_temp, temp_bytes);
  TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys, d_keys_sorted, d_vals, d_sorted_leaf, (int)n, 0, 63, st));
  hipLaunchKernelGGL(k_gather_boxes, dim3(grid_n), dim3(kBuildBlock), 0, st, d_boxes, d_sorted_leaf, n, d_sorted_boxes);
  TRY(hipGetLastError());
  uint32_t n_wide = 0;
  int need0 = 0;
  int mode = 0;
  uint32_t sah_run = 128;  // runs up to this length are cut by the surface-area heuristic (one thread sweeps a run)
  if (const char* e = std::getenv("RTAMD_BUILD_SAH_RUN")) sah_run = (uint32_t)std::max(0, atoi(e));  // measurement switch
  if (const char* e = std::getenv("RTAMD_BUILD_MODE")) mode = std::max(0, std::min(2, atoi(e)));  // measurement switch
  for (; mode < 3; ++mode) {
    const uint32_t one = 1u, zero = 0u;
    TRY(hipMemcpyAsync(d_n_wide, &one, 4, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_run_lo, &zero, 4, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_run_hi, &n, 4, hipMemcpyHostToDevice, st));
    std::vector<uint32_t> level_begin{0u};
    uint32_t begin = 0, end = 1;
    while (begin < end) {
      hipLaunchKernelGGL(k_split_level, dim3((end - begin + kBuildBlock - 1) / kBuildBlock), dim3(kBuildBlock), 0, st, begin, end, d_keys_sorted, d_sorted_leaf,
                         d_words, d_sorted_boxes, d_suffix, mode, sah_run, d_run_lo, d_run_hi, d_cuts, d_n_inner);
      size_t sb = temp_bytes;
      TRY(hipcub::DeviceScan::ExclusiveSum(d_temp, sb, d_n_inner, d_first_child, (int)(end - begin), st));
      hipLaunchKernelGGL(k_emit_level, dim3((end - begin + kBuildBlock - 1) / kBuildBlock), dim3(kBuildBlock), 0, st, begin, end, d_cuts, d_n_inner, d_first_child,
                         d_sorted_leaf, d_words, d_run_lo, d_run_hi, d_n_wide, d_wide);
      TRY(hipMemcpyAsync(&n_wide, d_n_wide, 4, hipMemcpyDeviceToHost, st));
      TRY(hipStreamSynchronize(st));
      begin = end; end = n_wide;
      level_begin.push_back(begin);
      if (level_begin.size() > 64) return bail(fail(RT_E_INVALID, "gpu_build_wide: internal error (tree deeper than 64 levels)"));
    }
    for (size_t lv = level_begin.size() - 1; lv-- > 0;) {
      const uint32_t b = level_begin[lv], e = level_begin[lv + 1];
      if (e > b) hipLaunchKernelGGL(k_boxes_level, dim3((e - b + kBuildBlock - 1) / kBuildBlock), dim3(kBuildBlock), 0, st, b, e, d_sorted_boxes, d_wide, d_node_box, d_need);
    }
    uint32_t need = 0;
    TRY(hipMemcpyAsync(&need, d_need, 4, hipMemcpyDeviceToHost, st));
    TRY(hipStreamSynchronize(st));
    TRY(hipGetLastError());
    need0 = (int)need;
    if (need <= kWideStack) break;  // otherwise once more, with cuts tied closer to the quartiles
  }
#undef TRY
#undef GET
  *wide_out = d_wide; *n_wide_out = n_wide; *stack_need_out = need0;
  return RT_OK;
}

}  // namespace rti
