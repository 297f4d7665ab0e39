// Kernels of preset PRESET_BOXES in float arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<float> kernel_table<float, rtd::PRESET_BOXES>() { return make_table<float, rtd::CfgBoxes>(); }
}  // namespace rti
