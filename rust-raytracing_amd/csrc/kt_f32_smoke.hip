// Kernels of preset PRESET_SMOKE in float arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<float> kernel_table<float, rtd::PRESET_SMOKE>() { return make_table<float, rtd::CfgSmoke>(); }
}  // namespace rti
