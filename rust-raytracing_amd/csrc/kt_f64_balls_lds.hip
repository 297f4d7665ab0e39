// Kernels of preset PRESET_BALLS_LDS in double arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_BALLS_LDS>() { return make_table<double, rtd::CfgBallsLds>(); }
}  // namespace rti
