// Single-precision kernels and their launch code (multiply-add contraction allowed).
#include "render_impl.h"
namespace rti {
RTI_INSTANTIATE(float)
}
