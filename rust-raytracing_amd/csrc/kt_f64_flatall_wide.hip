// Kernels of preset PRESET_FLATALL_WIDE in double arithmetic (the library's own tree in its four-wide form; every feature but nested trees).
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_FLATALL_WIDE>() { return make_table<double, rtd::CfgFlatAllWide>(); }
}  // namespace rti
