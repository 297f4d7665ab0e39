// Kernels of preset PRESET_FLATALL_LDS in float arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<float> kernel_table<float, rtd::PRESET_FLATALL_LDS>() { return make_table<float, rtd::CfgFlatAllLds>(); }
}  // namespace rti
