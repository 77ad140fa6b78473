// Instantiations: second generation of the 3x3 stride-1 convolution on rows of 12 / 14 / 16 pixels (ms_conv_k3n.h), prologue-free and BatchNorm-apply variants.
#include "ms_conv_k3n.h"
namespace ms {
int conv_dispatch_k3n_p2(const ConvArgs& a, int mt, hipStream_t st);      // ms_conv_inst_k3n2.hip: the two-tensor prologue
template <int W>
static int k3n_w(const ConvArgs& a, int mt, hipStream_t st) {
  if (a.pro_mode == 0) return mt == 2 ? launch_conv_k3n_t<W, 2, 0>(a, st) : launch_conv_k3n_t<W, 1, 0>(a, st);
  return mt == 2 ? launch_conv_k3n_t<W, 2, 1>(a, st) : launch_conv_k3n_t<W, 1, 1>(a, st);
}
int conv_dispatch_k3n(const ConvArgs& a, hipStream_t st) {
  const int mt = conv_k3n_mt(a);
  if (a.pro_mode == 2) return conv_dispatch_k3n_p2(a, mt, st);
  switch (a.Ws) {
    case 12: return k3n_w<12>(a, mt, st);
    case 14: return k3n_w<14>(a, mt, st);
    default: return k3n_w<16>(a, mt, st);
  }
}
}  // namespace ms
