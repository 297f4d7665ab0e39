// Shared between the translation units of librtamd.so: the context object, error plumbing and the
// per-precision entry points.  The f32 and f64 kernels live in separate translation units (rt_f32.hip,
// rt_f64.hip) so that the f64 build can be compiled without multiply-add contraction: its results then
// track a plain IEEE f64 evaluation (what the Rust reference does) operation for operation.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "kernel_table.h"
#include "rtamd.h"

namespace rti {
extern thread_local std::string g_err;
inline int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return ::rti::fail(RT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

constexpr uint32_t kBlock = 256;
constexpr uint32_t kLdsAllMax = 64u * 1024u;  // stage the whole scene in LDS up to this size
constexpr uint32_t kLdsTopMax = 32u * 1024u;  // otherwise stage this many bytes of top-level BVH links
constexpr uint64_t kTargetPathsPerPass = 64ull << 20;
constexpr uint32_t kMaxDepth = 4096;
constexpr uint32_t kRayCountWords = 16 + 8 * 4 * 4096;  // ray counter, then 8 diagnostic words per wave of up to 4096 workgroups
using rtd::kFlatMaxLeaves;  // scenes with at most this many leaves take the small-scene presets: every leaf tested, in tree order, no box tests
constexpr uint32_t kSegments = 64;  // queue segments (a multiple of the 8 XCDs; at most the workgroup size)
constexpr uint32_t kCounterWords = (kMaxDepth + 2) * kSegments * 11;  // per bounce and segment: queue count, 2 chunk cursors, 8 class counts

template <class R> struct SceneImage {
  uint8_t* d_blob = nullptr;
  double* d_exact_boxes = nullptr;  // DeviceScene::exact_boxes, when the image has them
  DeviceScene<R> dev{};
  int preset = PRESET_ALL_TOP;
  KernelTable<R> table{};
  uint32_t lds_bytes = 0;
  bool device_built_tree = false;  // its four-wide tree came from gpu_build_wide()
};

struct TmpLink {
  double bmin[3], bmax[3];
  uint32_t hit;
  int32_t miss;
  int32_t depth, tree;
};

// A node of the four-wide form of the library's own tree before it is packed for the device (device_scene.h: WideNode).
struct WideTmp {
  double lo[4][3], hi[4][3];  // [child][axis]
  uint32_t child[4];          // index of another node, a leaf word, or kWideNone
};

// A hit that does not write (u, v) - a moving sphere (src/moving_sphere.rs:44-72), a medium (src/constant_medium.rs:97-108) -
// reports whatever an earlier accepted candidate left in the reference's shared record.  Only an image texture reads (u, v):
// true when one sits on such an object.
inline bool reads_leftover_uv(const rt_scene_desc& s) {
  auto reads_uv = [&](int mat) {
    if (mat < 0 || mat >= s.n_materials) return false;
    const int tx = s.materials[mat].texture;
    if (tx < 0 || tx >= s.n_textures) return false;
    const rt_texture& t = s.textures[tx];
    auto image = [&](int k) { return k >= 0 && k < s.n_textures && s.textures[k].tag == RT_TEX_IMAGE; };
    return t.tag == RT_TEX_IMAGE || (t.tag == RT_TEX_CHECKER && (image(t.even) || image(t.odd)));
  };
  for (int i = 0; i < s.n_objects; ++i) {
    const rt_object& o = s.objects[i];
    if ((o.flags & RT_OBJ_MEDIUM) && reads_uv(o.medium_material)) return true;
    if (!(o.flags & RT_OBJ_SUB_BVH) && s.shapes[o.geom].tag == RT_SHAPE_MOVING_SPHERE && reads_uv(s.shapes[o.geom].material)) return true;
  }
  return false;
}

inline uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }

// Host threads this process may use for the upload's loops: the hardware's, capped by OMP_NUM_THREADS (a GPU box's CPU share is
// far below the machine's core count) and by 16.
inline unsigned upload_threads() {
  unsigned hw = std::thread::hardware_concurrency();
  if (const char* e = std::getenv("OMP_NUM_THREADS")) hw = std::min<unsigned>(hw ? hw : 1u, (unsigned)std::max(1, atoi(e)));
  return std::max(1u, std::min(hw, 16u));
}
// fn(thread, lo, hi) over [0, n) in contiguous chunks, one per thread, in order (thread t gets the t-th chunk); small ranges run inline
template <class Fn> void parallel_chunks(size_t n, Fn fn, size_t min_per_thread = 32768) {
  const size_t n_threads = std::max<size_t>(1, std::min<size_t>(upload_threads(), n / std::max<size_t>(1, min_per_thread)));
  if (n_threads <= 1) { fn((size_t)0, (size_t)0, n); return; }
  const size_t per = (n + n_threads - 1) / n_threads;
  std::vector<std::thread> pool;
  for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(fn, t, std::min(n, t * per), std::min(n, (t + 1) * per));
  fn((size_t)0, (size_t)0, std::min(n, per));
  for (std::thread& t : pool) t.join();
}
inline size_t parallel_chunk_count(size_t n, size_t min_per_thread = 32768) {
  return std::max<size_t>(1, std::min<size_t>(upload_threads(), n / std::max<size_t>(1, min_per_thread)));
}

}  // namespace rti

struct rt_context {
  int device = 0;
  hipStream_t stream = nullptr;
  int n_cus = 256;
  bool has_scene = false;
  rti::SceneImage<float> sf;
  rti::SceneImage<double> sd;
  rti::SceneImage<double> sd_own;  // double-precision image over the library's own tree (RT_F64_OWN_TREE); empty for small-scene presets
  bool own64 = false;              // the call in progress asked for it
  rtd::DImage* d_images = nullptr;
  std::vector<uint8_t*> d_image_bytes;
  // render workspace
  void* ws = nullptr;
  size_t ws_bytes = 0;
  uint32_t* d_counters = nullptr;  // [kCounterWords]
  unsigned long long* d_ray_count = nullptr;
  void* accum = nullptr;
  size_t accum_bytes = 0;
  uint8_t* d_rgb = nullptr;  // rt_resolve: the RGB8 image on the device, before its copy to the host
  size_t rgb_bytes = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long* h_rays = nullptr;  // one page-locked host word: where the allocation-free entry point receives its ray count
  // progressive rendering (rt_progress_*): the frame in progress and the next sample to render
  rt_render_params progress{};
  uint32_t progress_next = 0;
  bool progress_active = false;
  // multi-GPU (comm.hip): RCCL communicator this context's GPU belongs to, its rank, and the packed W*H*3 staging buffer
  void* comm = nullptr;
  int comm_rank = 0, comm_size = 1;
  void* comm_buf = nullptr;
  size_t comm_buf_bytes = 0;
  void* scratch = nullptr;               // this device's address of the library's static scratch array (see rtamd_api.hip)
};

namespace rti {

template <class R> SceneImage<R>& image_of(rt_context* c);
template <> inline SceneImage<float>& image_of<float>(rt_context* c) { return c->sf; }
template <> inline SceneImage<double>& image_of<double>(rt_context* c) { return (c->own64 && c->sd_own.d_blob) ? c->sd_own : c->sd; }
// Strips RT_F64_OWN_TREE from a precision argument and notes on the context whether the call asked for the library's own tree.
inline int select_tree(rt_context* c, int precision) {
  if (c) c->own64 = (precision & RT_F64_OWN_TREE) != 0 && (precision & ~RT_F64_OWN_TREE) == RT_PRECISION_F64;
  return precision & ~RT_F64_OWN_TREE;
}

// Workgroups per CU that stay resident for a kernel: the runtime's estimate, at most 8 (two waves per SIMD and workgroup), and one
// less than the estimate when that leaves a margin (the estimate has been seen one too high for kernels near an SGPR allocation
// step).  The kernels take their work from cursors, so a grid above or below the true residency is only a small inefficiency.
inline uint32_t resident_blocks(int api_estimate) {
  if (const char* e = getenv("RTAMD_BLOCKS_PER_CU")) return (uint32_t)std::max(1, atoi(e));
  if (api_estimate <= 0) return 4u;
  return (uint32_t)std::min(api_estimate, 8);
}

inline uint32_t grid_for(const rt_context* c, uint64_t n, uint32_t blocks_per_cu) {
  uint64_t want = (n + kBlock - 1) / kBlock;
  uint64_t cap = (uint64_t)c->n_cus * blocks_per_cu;
  uint64_t g = std::max<uint64_t>(1, std::min(want, cap));
  return (uint32_t)((g + kSegments - 1) / kSegments * kSegments);  // whole workgroups per queue segment
}

// Implemented once per precision (render_impl.hip).
template <class R> void layout_sections(rtd::BlobLayout& lay);
template <class R>
int build_image(rt_context* c, const rt_scene_desc& s, const std::vector<TmpLink>& links, int root_link, const std::vector<int>& sub_root_of_object,
                int preset, SceneImage<R>& im, const std::vector<WideTmp>* wide, const std::vector<uint32_t>* pre, bool exact,
                const rtd::WideNode* device_wide = nullptr, uint32_t n_device_wide = 0);
// gpu_build.hip: (key, value) pairs in host memory sorted by key on the device (stable); for the upload's searches among a million hashes
int gpu_sort_pairs(rt_context* c, uint64_t* keys, uint32_t* vals, size_t n);
// gpu_build.hip: the four-wide tree over `n_leaves` boxes built on the device (a device array the caller frees)
int gpu_build_wide(rt_context* c, const float* boxes, const uint32_t* words, size_t n_leaves, const double cen_lo[3], const double cen_hi[3],
                   rtd::WideNode** wide_out, uint32_t* n_wide_out, int* stack_need_out);
template <class R>
int render_accum_impl(rt_context* c, const rt_render_params& p, void* accum_dev, rt_stats* stats, unsigned long long* counters_override = nullptr,
                      unsigned long long* host_rays = nullptr);
template <class R> int trace_batch_impl(rt_context* c, const rt_ray* rays, size_t n, rt_hit* hits, uint64_t seed, int mode);
template <class R>
int pairs_kernel_impl(rt_context* c, int which, uint32_t W, uint32_t H, uint32_t max_depth, const uint32_t* pixel, const uint32_t* sample, size_t n,
                      uint64_t seed, void* out, size_t out_bytes);
template <class R> int resolve_impl(rt_context* c, const void* accum_dev, size_t n_pixels, uint32_t spp, uint8_t* d_rgb);

}  // namespace rti
