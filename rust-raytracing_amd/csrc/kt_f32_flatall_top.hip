// Kernels of preset PRESET_FLATALL_TOP in float arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<float> kernel_table<float, rtd::PRESET_FLATALL_TOP>() { return make_table<float, rtd::CfgFlatAllTop>(); }
}  // namespace rti
