// One table of kernel launchers per (arithmetic type, preset).  Each table is instantiated in a translation unit of its
// own (kt_*.hip) so the presets compile in parallel and the f64 ones can be built without multiply-add contraction.
#pragma once
#include "kernels.h"
#include "rtamd.h"

namespace rti {
using namespace rtd;

template <class R> struct KernelTable {
  uint32_t (*lds_bytes)(const BlobLayout& lay, uint32_t lds_links);
  void (*bounce_fused)(uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, const FrameParams& fp, PathQueue<R> in, PathQueue<R> out,
                       uint32_t* counters, uint32_t* cursors, unsigned long long* ray_count, R* accum, uint32_t bounce);
  // returns which instantiation it launched: RT_VARIANT_* bits (rtamd.h: rt_stats.kernel_variant)
  uint32_t (*paths_persistent)(uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, const FrameParams& fp, unsigned long long total_paths,
                               unsigned long long* path_cursor, unsigned long long* ray_count, R* accum);
  int (*resident_blocks_per_cu)(int kernel, uint32_t lds);  // kernel: 0 bounce_fused, 1 trace, 2 shade, 3 paths_persistent
  void (*trace)(uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, const FrameParams& fp, PathQueue<R> in, HitQueue<R> hits,
                uint32_t* counters, uint32_t* cursors, unsigned long long* ray_count, uint32_t* class_counters, uint32_t* class_index, uint32_t capacity, R* accum,
                uint32_t bounce);
  void (*shade)(uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, const FrameParams& fp, PathQueue<R> in, HitQueue<R> hits,
                PathQueue<R> out, uint32_t* counters, uint32_t* cursors, const uint32_t* class_counters, const uint32_t* class_index, uint32_t capacity,
                R* accum, uint32_t bounce);
  void (*first_hit)(uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, const RayIO* rays, uint32_t n, HitIO* hits, uint64_t seed,
                    int mode);
  void (*sample_radiance)(uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, uint32_t W, uint32_t H, uint32_t max_depth,
                          const uint32_t* pixel, const uint32_t* sample, uint32_t n, uint64_t seed, double* out, double* path_out, uint32_t max_vertices);
};

constexpr uint32_t kBlockThreads = 256;

// Per-lane walks use the stepwise kernel, except the presets that carry nested trees: their walk state does not fit the
// register file at a useful occupancy, and the monolithic kernel is the faster of the two there.
#ifndef RT_STEPWISE_NESTED
#define RT_STEPWISE_NESTED 0
#endif
// The chunked variant of the persistent kernel exists for the small sphere scenes only (see render_accum_impl for when it runs).
template <class C> constexpr bool chunked_variant() { return C::walk == WALK_UNIFORM && C::has(F_SPHERE); }
// Double precision on the caller's tree: the stepwise kernel has a second instantiation that builds its records after the walk,
// launched for scenes in which no texture reads a leftover (u, v) (DeviceScene::defer_ok).
template <class R, class C> constexpr bool list_defer_variant() {
  return sizeof(R) == 8 && C::walk != WALK_UNIFORM && C::walk != WALK_LANE_WIDE && !C::has(F_NESTED) && C::costly_records();
}
template <class C> constexpr bool use_stepwise() { return C::walk != WALK_UNIFORM && (RT_STEPWISE_NESTED || !C::has(F_NESTED)); }

template <class R, class C> KernelTable<R> make_table() {
  KernelTable<R> t;
  t.lds_bytes = [](const BlobLayout& lay, uint32_t lds_links) { return lds_bytes_for<R, C>(lay, lds_links); };
  t.bounce_fused = [](uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, const FrameParams& fp, PathQueue<R> in, PathQueue<R> out,
                      uint32_t* counters, uint32_t* cursors, unsigned long long* ray_count, R* accum, uint32_t bounce) {
    hipLaunchKernelGGL((k_bounce_fused<R, C>), dim3(grid), dim3(kBlockThreads), lds, st, sc, fp, in, out, counters, cursors, ray_count, accum, bounce);
  };
  t.paths_persistent = [](uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, const FrameParams& fp, unsigned long long total_paths,
                          unsigned long long* path_cursor, unsigned long long* ray_count, R* accum) -> uint32_t {
    const bool fixed = fp.path_grab == kPathGrab;
    const uint32_t vf = fixed ? RT_VARIANT_FIXED_GRAB : 0u;
    if constexpr (!use_stepwise<C>()) {
      if constexpr (chunked_variant<C>()) {
        if (fp.chunk > 1u) {
          if (fixed) hipLaunchKernelGGL((k_paths_persistent<R, C, true, true>), dim3(grid), dim3(kBlockThreads), lds, st, sc, fp, total_paths, path_cursor, ray_count, accum);
          else hipLaunchKernelGGL((k_paths_persistent<R, C, false, true>), dim3(grid), dim3(kBlockThreads), lds, st, sc, fp, total_paths, path_cursor, ray_count, accum);
          return vf | RT_VARIANT_CHUNKED;
        }
      }
      if (fixed) hipLaunchKernelGGL((k_paths_persistent<R, C, true>), dim3(grid), dim3(kBlockThreads), lds, st, sc, fp, total_paths, path_cursor, ray_count, accum);
      else hipLaunchKernelGGL((k_paths_persistent<R, C, false>), dim3(grid), dim3(kBlockThreads), lds, st, sc, fp, total_paths, path_cursor, ray_count, accum);
      return vf;
    }
    else {
      if constexpr (list_defer_variant<R, C>()) {
        if (sc.defer_ok) {
          if (fixed) hipLaunchKernelGGL((k_paths_stepwise<R, C, true, true>), dim3(grid), dim3(kStepThreads), lds, st, sc, fp, total_paths, path_cursor, ray_count, accum);
          else hipLaunchKernelGGL((k_paths_stepwise<R, C, false, true>), dim3(grid), dim3(kStepThreads), lds, st, sc, fp, total_paths, path_cursor, ray_count, accum);
          return vf | RT_VARIANT_STEPWISE | RT_VARIANT_DEFERRED_RECORDS;
        }
      }
      if (fixed) hipLaunchKernelGGL((k_paths_stepwise<R, C, true>), dim3(grid), dim3(kStepThreads), lds, st, sc, fp, total_paths, path_cursor, ray_count, accum);
      else hipLaunchKernelGGL((k_paths_stepwise<R, C, false>), dim3(grid), dim3(kStepThreads), lds, st, sc, fp, total_paths, path_cursor, ray_count, accum);
      return vf | RT_VARIANT_STEPWISE;
    }
  };
  t.resident_blocks_per_cu = [](int kernel, uint32_t lds) {
    int n = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (kernel) {
      case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bounce_fused<R, C>, (int)kBlockThreads, lds); break;
      case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace<R, C>, (int)kBlockThreads, lds); break;
      case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_shade<R, C>, (int)kBlockThreads, lds); break;
      case 3:
        if constexpr (!use_stepwise<C>()) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_paths_persistent<R, C, true>, (int)kBlockThreads, lds);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_paths_stepwise<R, C, true>, (int)kStepThreads, lds);
        break;
      default: break;
    }
    return e == hipSuccess ? n : -1;
  };
  t.trace = [](uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, const FrameParams& fp, PathQueue<R> in, HitQueue<R> hits,
               uint32_t* counters, uint32_t* cursors, unsigned long long* ray_count, uint32_t* class_counters, uint32_t* class_index, uint32_t capacity,
               R* accum, uint32_t bounce) {
    hipLaunchKernelGGL((k_trace<R, C>), dim3(grid), dim3(kBlockThreads), lds, st, sc, fp, in, hits, counters, cursors, ray_count, class_counters, class_index,
                       capacity, accum, bounce);
  };
  t.shade = [](uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, const FrameParams& fp, PathQueue<R> in, HitQueue<R> hits,
               PathQueue<R> out, uint32_t* counters, uint32_t* cursors, const uint32_t* class_counters, const uint32_t* class_index, uint32_t capacity,
               R* accum, uint32_t bounce) {
    hipLaunchKernelGGL((k_shade<R, C>), dim3(grid), dim3(kBlockThreads), lds, st, sc, fp, in, hits, out, counters, cursors, class_counters, class_index,
                       capacity, accum, bounce);
  };
  t.first_hit = [](uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, const RayIO* rays, uint32_t n, HitIO* hits, uint64_t seed,
                   int mode) {
    if (mode == 2) hipLaunchKernelGGL((k_first_hit<R, C, 2>), dim3(grid), dim3(kBlockThreads), lds, st, sc, rays, n, hits, seed);
    else if (mode == 1) hipLaunchKernelGGL((k_first_hit<R, C, 1>), dim3(grid), dim3(kBlockThreads), lds, st, sc, rays, n, hits, seed);
    else hipLaunchKernelGGL((k_first_hit<R, C, 0>), dim3(grid), dim3(kBlockThreads), lds, st, sc, rays, n, hits, seed);
  };
  t.sample_radiance = [](uint32_t grid, uint32_t lds, hipStream_t st, const DeviceScene<R>& sc, uint32_t W, uint32_t H, uint32_t max_depth,
                         const uint32_t* pixel, const uint32_t* sample, uint32_t n, uint64_t seed, double* out, double* path_out,
                         uint32_t max_vertices) {
    hipLaunchKernelGGL((k_sample_radiance<R, C>), dim3(grid), dim3(kBlockThreads), lds, st, sc, W, H, max_depth, pixel, sample, n, seed, out, path_out,
                       max_vertices);
  };
  return t;
}

// Defined once per (R, preset) in kt_*.hip.
template <class R, int PRESET> KernelTable<R> kernel_table();

// Simple kernels without a preset (one instantiation per arithmetic type, in kt_*_common.hip).
template <class R> void launch_gen_primary(uint32_t grid, hipStream_t st, const DeviceScene<R>& sc, const FrameParams& fp, PathQueue<R> out, uint32_t* counters);
template <class R> void launch_resolve(uint32_t grid, hipStream_t st, const R* accum, uint32_t n_pixels, uint32_t spp, uint8_t* rgb);
template <class R>
void launch_camera_rays(uint32_t grid, hipStream_t st, const DeviceScene<R>& sc, uint32_t W, uint32_t H, const uint32_t* pixel, const uint32_t* sample,
                        uint32_t n, uint64_t seed, RayIO* out);

}  // namespace rti
