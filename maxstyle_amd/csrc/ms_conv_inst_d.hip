// Instantiations: stride-2 convolutions (3x3 p1 down-sampling; 2x2 p0 = data-gradient of ConvTranspose2d k2 s2).
#include "ms_conv_kernel.h"
namespace ms {
template <int KS, bool VEC, bool NARROW>
static int s2_nt(const ConvArgs& a, int nt, hipStream_t st) {
  switch (nt) {
    case 1: return launch_conv<KS, 2, FETCH_NORMAL, 1, VEC, NARROW, false>(a, st);
    case 2: return launch_conv<KS, 2, FETCH_NORMAL, 2, VEC, NARROW, false>(a, st);
    default: return launch_conv<KS, 2, FETCH_NORMAL, 4, VEC, NARROW, false>(a, st);
  }
}
template <int KS>
static int s2_ks(const ConvArgs& a, int nt, bool vec, bool narrow, hipStream_t st) {
  if (vec) return narrow ? s2_nt<KS, true, true>(a, nt, st) : s2_nt<KS, true, false>(a, nt, st);
  return narrow ? s2_nt<KS, false, true>(a, nt, st) : s2_nt<KS, false, false>(a, nt, st);
}
int conv_dispatch_s2(const ConvArgs& a, int ks, int nt, bool vec, bool narrow, hipStream_t st) {
  return ks == 3 ? s2_ks<3>(a, nt, vec, narrow, st) : s2_ks<2>(a, nt, vec, narrow, st);
}
}  // namespace ms
