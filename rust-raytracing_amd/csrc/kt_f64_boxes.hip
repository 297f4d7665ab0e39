// Kernels of preset PRESET_BOXES in double arithmetic.
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_BOXES>() { return make_table<double, rtd::CfgBoxes>(); }
}  // namespace rti
