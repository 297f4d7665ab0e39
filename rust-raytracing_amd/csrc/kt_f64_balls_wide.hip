// Kernels of preset PRESET_BALLS_WIDE in double arithmetic (the library's own tree in its four-wide form; sphere scenes).
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_BALLS_WIDE>() { return make_table<double, rtd::CfgBallsWide>(); }
}  // namespace rti
