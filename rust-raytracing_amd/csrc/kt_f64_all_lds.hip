// Kernels of preset PRESET_ALL_LDS in double arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_ALL_LDS>() { return make_table<double, rtd::CfgAllLds>(); }
}  // namespace rti
