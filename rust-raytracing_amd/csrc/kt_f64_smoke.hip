// Kernels of preset PRESET_SMOKE in double arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_SMOKE>() { return make_table<double, rtd::CfgSmoke>(); }
}  // namespace rti
