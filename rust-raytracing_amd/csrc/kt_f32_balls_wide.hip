// Kernels of preset PRESET_BALLS_WIDE in float arithmetic (the library's own tree in its four-wide form; sphere scenes).
#include "kernel_table.h"
namespace rti {
template <> KernelTable<float> kernel_table<float, rtd::PRESET_BALLS_WIDE>() { return make_table<float, rtd::CfgBallsWide>(); }
}  // namespace rti
