// Kernels of preset PRESET_ALL_TOP in double arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_ALL_TOP>() { return make_table<double, rtd::CfgAllTop>(); }
}  // namespace rti
