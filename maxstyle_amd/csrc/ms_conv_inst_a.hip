// Instantiations: 3x3 stride-1 convolutions, plain fetch (forward convs and stride-1 data-gradients).
#include "ms_conv_kernel.h"
namespace ms {
template <bool VEC, bool NARROW, bool IN2>
static int k3s1_nt(const ConvArgs& a, int nt, hipStream_t st) {
  switch (nt) {
    case 1: return launch_conv<3, 1, FETCH_NORMAL, 1, VEC, NARROW, IN2>(a, st);
    case 2: return launch_conv<3, 1, FETCH_NORMAL, 2, VEC, NARROW, IN2>(a, st);
    default: return launch_conv<3, 1, FETCH_NORMAL, 4, VEC, NARROW, IN2>(a, st);
  }
}
int conv_dispatch_k3s1_plain(const ConvArgs& a, int nt, bool vec, bool narrow, bool in2, hipStream_t st) {
  if (vec) {
    if (narrow) return in2 ? k3s1_nt<true, true, true>(a, nt, st) : k3s1_nt<true, true, false>(a, nt, st);
    return in2 ? k3s1_nt<true, false, true>(a, nt, st) : k3s1_nt<true, false, false>(a, nt, st);
  }
  if (narrow) return in2 ? k3s1_nt<false, true, true>(a, nt, st) : k3s1_nt<false, true, false>(a, nt, st);
  return in2 ? k3s1_nt<false, false, true>(a, nt, st) : k3s1_nt<false, false, false>(a, nt, st);
}
}  // namespace ms
