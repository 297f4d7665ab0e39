// Kernels of preset PRESET_BALLS_TOP in float arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<float> kernel_table<float, rtd::PRESET_BALLS_TOP>() { return make_table<float, rtd::CfgBallsTop>(); }
}  // namespace rti
