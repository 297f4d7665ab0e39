// Kernels of preset PRESET_ALL_TOP in float arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<float> kernel_table<float, rtd::PRESET_ALL_TOP>() { return make_table<float, rtd::CfgAllTop>(); }
}  // namespace rti
