// Instantiations: the streaming 1x1 convolution kernel (ms_conv_k1s.h).
#include "ms_conv_k1s.h"
namespace ms {
template <int NT, int EPI>
static int k1s_ncg(const ConvArgs& a, hipStream_t st) {
  const int ncg = a.cin_pad / 4;
  if (ncg == 4) return launch_conv_k1s_t<NT, EPI, 4>(a, st);
  if (ncg == 8 || kK1sD == 8) return launch_conv_k1s_t<NT, EPI, 8>(a, st);
  return launch_conv_k1s_t<NT, EPI, 16>(a, st);
}
template <int NT>
static int k1s_epi(const ConvArgs& a, hipStream_t st) {
  if (a.epi_mode == 4) return k1s_ncg<NT, 4>(a, st);
  if (a.epi_mode == 5) return k1s_ncg<NT, 5>(a, st);
  return k1s_ncg<NT, 0>(a, st);
}
int conv_dispatch_k1s(const ConvArgs& a, hipStream_t st) {
  if (a.epi_mode == 2) return k1s_ncg<4, 2>(a, st);
  if (a.Cout <= 16) return k1s_epi<1>(a, st);
  if (a.Cout <= 32) return k1s_epi<2>(a, st);
  return k1s_epi<4>(a, st);
}
}  // namespace ms
