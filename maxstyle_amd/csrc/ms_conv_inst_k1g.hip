// Instantiations: LDS-tiled GEMM form of the channel-heavy 1x1 convolutions (ms_conv_k1g.h).
#include "ms_conv_k1g.h"
namespace ms {
template <int NT>
static int k1g_epi(const ConvArgs& a, hipStream_t st) {
  if (a.epi_mode == 5) return launch_conv_k1g_t<NT, 5>(a, st);
  return a.epi_mode == 4 ? launch_conv_k1g_t<NT, 4>(a, st) : launch_conv_k1g_t<NT, 0>(a, st);
}
int conv_dispatch_k1g(const ConvArgs& a, hipStream_t st) {
  const int nt = conv_k1g_nt(a);
  return nt == 4 ? k1g_epi<4>(a, st) : (nt == 2 ? k1g_epi<2>(a, st) : k1g_epi<1>(a, st));
}
}  // namespace ms
