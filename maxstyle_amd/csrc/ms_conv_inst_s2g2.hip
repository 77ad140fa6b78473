// Instantiations: second generation of the 3x3 stride-2 forward convolution (ms_conv_s2.h).
#include "ms_conv_s2.h"
namespace ms {
template <int GEO>
static int s2g2_nt(const ConvArgs& a, int nt, hipStream_t st) {
  if (nt == 4) return launch_conv_s2_t<GEO, 4>(a, st);
  if (nt == 2) return launch_conv_s2_t<GEO, 2>(a, st);
  return launch_conv_s2_t<GEO, 1>(a, st);
}
int conv_dispatch_s2g2(const ConvArgs& a, hipStream_t st) {
  int geo, nt;
  conv_s2g2_plan(a, geo, nt);
  return geo ? s2g2_nt<1>(a, nt, st) : s2g2_nt<0>(a, nt, st);
}
}  // namespace ms
