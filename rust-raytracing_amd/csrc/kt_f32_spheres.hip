// Kernels of preset PRESET_SPHERES in float arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<float> kernel_table<float, rtd::PRESET_SPHERES>() { return make_table<float, rtd::CfgSpheres>(); }
}  // namespace rti
