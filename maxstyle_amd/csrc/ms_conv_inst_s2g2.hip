// Instantiations: second generation of the 3x3 stride-2 forward convolution (ms_conv_s2.h).
#include "ms_conv_s2.h"
namespace ms {
template <int GEO, int PRO>
static int s2g2_nt(const ConvArgs& a, int nt, hipStream_t st) {
  if (nt == 4) return launch_conv_s2_t<GEO, 4, PRO>(a, st);
  if (nt == 2) return launch_conv_s2_t<GEO, 2, PRO>(a, st);
  return launch_conv_s2_t<GEO, 1, PRO>(a, st);
}
int conv_dispatch_s2g2(const ConvArgs& a, hipStream_t st) {
  int geo, nt;
  conv_s2g2_plan(a, geo, nt);
  if (a.pro_mode == 1) return s2g2_nt<0, 1>(a, nt, st);      // (eligible only with the tile geometry)
  return geo ? s2g2_nt<1, 0>(a, nt, st) : s2g2_nt<0, 0>(a, nt, st);
}
}  // namespace ms
