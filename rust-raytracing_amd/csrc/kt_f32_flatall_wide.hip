// Kernels of preset PRESET_FLATALL_WIDE in float arithmetic (the library's own tree in its four-wide form; every feature but nested trees).
#include "kernel_table.h"
namespace rti {
template <> KernelTable<float> kernel_table<float, rtd::PRESET_FLATALL_WIDE>() { return make_table<float, rtd::CfgFlatAllWide>(); }
}  // namespace rti
