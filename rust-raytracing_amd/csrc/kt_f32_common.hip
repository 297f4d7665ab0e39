// Preset-independent kernels in float arithmetic: primary rays, resolve, camera-ray inspection.
#include "kernel_table.h"
namespace rti {
template <> void launch_gen_primary<float>(uint32_t grid, hipStream_t st, const DeviceScene<float>& sc, const FrameParams& fp, PathQueue<float> out, uint32_t* counters) {
  hipLaunchKernelGGL((k_gen_primary<float>), dim3(grid), dim3(kBlockThreads), 0, st, sc, fp, out, counters);
}
template <> void launch_resolve<float>(uint32_t grid, hipStream_t st, const float* accum, uint32_t n_pixels, uint32_t spp, uint8_t* rgb) {
  hipLaunchKernelGGL((k_resolve<float>), dim3(grid), dim3(kBlockThreads), 0, st, accum, n_pixels, spp, rgb);
}
template <>
void launch_camera_rays<float>(uint32_t grid, hipStream_t st, const DeviceScene<float>& sc, uint32_t W, uint32_t H, const uint32_t* pixel,
                               const uint32_t* sample, uint32_t n, uint64_t seed, RayIO* out) {
  hipLaunchKernelGGL((k_camera_rays<float>), dim3(grid), dim3(kBlockThreads), 0, st, sc, W, H, pixel, sample, n, seed, out);
}
}  // namespace rti
