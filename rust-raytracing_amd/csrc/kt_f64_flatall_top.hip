// Kernels of preset PRESET_FLATALL_TOP in double arithmetic (the library's own tree, nested trees dissolved: RT_F64_OWN_TREE).
#include "kernel_table.h"
namespace rti {
template <> KernelTable<double> kernel_table<double, rtd::PRESET_FLATALL_TOP>() { return make_table<double, rtd::CfgFlatAllTop>(); }
}  // namespace rti
