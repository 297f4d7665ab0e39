// Kernels of preset PRESET_BALLS_TOP in double arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_BALLS_TOP>() { return make_table<double, rtd::CfgBallsTop>(); }
}  // namespace rti
