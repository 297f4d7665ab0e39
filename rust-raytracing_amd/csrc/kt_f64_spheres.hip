// Kernels of preset PRESET_SPHERES in double arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_SPHERES>() { return make_table<double, rtd::CfgSpheres>(); }
}  // namespace rti
