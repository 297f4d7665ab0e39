// Kernels of preset PRESET_BALLS_LDS in float arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<float> kernel_table<float, rtd::PRESET_BALLS_LDS>() { return make_table<float, rtd::CfgBallsLds>(); }
}  // namespace rti
