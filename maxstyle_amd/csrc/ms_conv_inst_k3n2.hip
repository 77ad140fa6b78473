// Instantiations: second generation of the 3x3 stride-1 convolution on rows of 12 / 14 / 16 pixels (ms_conv_k3n.h), BatchNorm-backward (two-tensor) prologue.
#include "ms_conv_k3n.h"
namespace ms {
int conv_dispatch_k3n_p2(const ConvArgs& a, int mt, hipStream_t st) {
  switch (a.Ws) {
    case 12: return mt == 2 ? launch_conv_k3n_t<12, 2, 2>(a, st) : launch_conv_k3n_t<12, 1, 2>(a, st);
    case 14: return mt == 2 ? launch_conv_k3n_t<14, 2, 2>(a, st) : launch_conv_k3n_t<14, 1, 2>(a, st);
    default: return mt == 2 ? launch_conv_k3n_t<16, 2, 2>(a, st) : launch_conv_k3n_t<16, 1, 2>(a, st);
  }
}
int conv_dispatch_k3n_s2(const ConvArgs& a, int mt, hipStream_t st) {
  switch (a.Wout) {
    case 12: return mt == 2 ? launch_conv_k3n_t<12, 2, 0, 3, 2>(a, st) : launch_conv_k3n_t<12, 1, 0, 3, 2>(a, st);
    case 14: return mt == 2 ? launch_conv_k3n_t<14, 2, 0, 3, 2>(a, st) : launch_conv_k3n_t<14, 1, 0, 3, 2>(a, st);
    default: return mt == 2 ? launch_conv_k3n_t<16, 2, 0, 3, 2>(a, st) : launch_conv_k3n_t<16, 1, 0, 3, 2>(a, st);
  }
}
int conv_dispatch_k3n_k1(const ConvArgs& a, int mt, hipStream_t st) {
  if (a.pro_mode == 2) return mt == 2 ? launch_conv_k3n_t<14, 2, 2, 1>(a, st) : launch_conv_k3n_t<14, 1, 2, 1>(a, st);
  if (a.pro_mode == 1) return mt == 2 ? launch_conv_k3n_t<14, 2, 1, 1>(a, st) : launch_conv_k3n_t<14, 1, 1, 1>(a, st);
  return mt == 2 ? launch_conv_k3n_t<14, 2, 0, 1>(a, st) : launch_conv_k3n_t<14, 1, 0, 1>(a, st);
}
}  // namespace ms
