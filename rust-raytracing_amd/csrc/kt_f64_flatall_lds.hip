// Kernels of preset PRESET_FLATALL_LDS in double arithmetic (the library's own tree, nested trees dissolved: RT_F64_OWN_TREE).
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_FLATALL_LDS>() { return make_table<double, rtd::CfgFlatAllLds>(); }
}  // namespace rti
