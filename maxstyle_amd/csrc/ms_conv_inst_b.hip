// Instantiations: 3x3 stride-1 convolutions with fused nearest up-sampling / zero-insertion fetch.
#include "ms_conv_kernel.h"
namespace ms {
int conv_dispatch_k3s1_plain(const ConvArgs& a, int nt, bool vec, bool narrow, bool in2, hipStream_t st);
template <int FETCH, bool NARROW>
static int k3s1f_nt(const ConvArgs& a, int nt, hipStream_t st) {
  switch (nt) {
    case 1: return launch_conv<3, 1, FETCH, 1, false, NARROW, false>(a, st);
    case 2: return launch_conv<3, 1, FETCH, 2, false, NARROW, false>(a, st);
    default: return launch_conv<3, 1, FETCH, 4, false, NARROW, false>(a, st);
  }
}
int conv_dispatch_k3s1(const ConvArgs& a, int fetch, int nt, bool vec, bool narrow, bool in2, hipStream_t st) {
  if (fetch == FETCH_UPS2) return narrow ? k3s1f_nt<FETCH_UPS2, true>(a, nt, st) : k3s1f_nt<FETCH_UPS2, false>(a, nt, st);
  if (fetch == FETCH_ZINS2) return narrow ? k3s1f_nt<FETCH_ZINS2, true>(a, nt, st) : k3s1f_nt<FETCH_ZINS2, false>(a, nt, st);
  return conv_dispatch_k3s1_plain(a, nt, vec, narrow, in2, st);
}
}  // namespace ms
