// Double-precision kernels and their launch code.  Compiled with -ffp-contract=off: the reference is plain IEEE f64
// (Rust never fuses a*b+c), and without contraction this build reproduces a CPU f64 evaluation operation for operation.
#include "render_impl.h"
namespace rti {
RTI_INSTANTIATE(double)
}
