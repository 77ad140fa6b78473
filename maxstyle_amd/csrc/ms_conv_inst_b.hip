// Instantiations: 3x3 stride-1 convolutions with fused nearest up-sampling / zero-insertion fetch.
#include "ms_conv_kernel.h"
namespace ms {
int conv_dispatch_k3s1_plain(const ConvArgs& a, int nt, bool vec, bool narrow, bool in2, hipStream_t st);
template <int FETCH, bool VEC, bool NARROW>
static int k3s1f_nt(const ConvArgs& a, int nt, hipStream_t st) {
  switch (nt) {
    case 1: return launch_conv<3, 1, FETCH, 1, VEC, NARROW, false>(a, st);
    default: return launch_conv<3, 1, FETCH, 2, VEC, NARROW, false>(a, st);
  }
}
template <int FETCH>
static int k3s1f(const ConvArgs& a, int nt, bool vec, bool narrow, hipStream_t st) {
  if (vec) return narrow ? k3s1f_nt<FETCH, true, true>(a, nt, st) : k3s1f_nt<FETCH, true, false>(a, nt, st);
  return narrow ? k3s1f_nt<FETCH, false, true>(a, nt, st) : k3s1f_nt<FETCH, false, false>(a, nt, st);
}
int conv_dispatch_k3s1(const ConvArgs& a, int fetch, int nt, bool vec, bool narrow, bool in2, hipStream_t st) {
  // fused-fetch variants: "vec" = 8-byte loads of the stored tensor expanded into the logical tile in LDS (needs an even stored width)
  const bool vec2 = (a.Ws % 2 == 0) && aligned16(a.in);
  if (fetch == FETCH_UPS2) return k3s1f<FETCH_UPS2>(a, nt > 2 ? 2 : nt, vec2, narrow, st);
  if (fetch == FETCH_ZINS2) return k3s1f<FETCH_ZINS2>(a, nt > 2 ? 2 : nt, vec2, narrow, st);
  return conv_dispatch_k3s1_plain(a, nt, vec, narrow, in2, st);
}
}  // namespace ms
