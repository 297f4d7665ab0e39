// Kernels of preset PRESET_FLATALL_WIDE_EXACT in double arithmetic (the library's own tree in its four-wide form, leaf tests and tie rule of the reference).
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_FLATALL_WIDE_EXACT>() { return make_table<double, rtd::CfgFlatAllWideExact>(); }
}  // namespace rti
