// Kernels of preset PRESET_ALL_LDS in float arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<float> kernel_table<float, rtd::PRESET_ALL_LDS>() { return make_table<float, rtd::CfgAllLds>(); }
}  // namespace rti
