// Instantiations: wide-read 3x3 stride-1 convolution (ms_conv_wide.h), 8-row tiles (two output rows per MFMA wave).
#include "ms_conv_wide.h"
namespace ms {
template <int NT>
static int wide_pro8(const ConvArgs& a, hipStream_t st) {
  switch (a.pro_mode) {
    case 0: return launch_conv_wide_r<NT, 0, 2>(a, st);
    case 1: return launch_conv_wide_r<NT, 1, 2>(a, st);
    default: return launch_conv_wide_r<NT, 2, 2>(a, st);
  }
}
int conv_dispatch_wide8(const ConvArgs& a, int nt, hipStream_t st) {
  return nt >= 2 ? wide_pro8<2>(a, st) : wide_pro8<1>(a, st);
}
}  // namespace ms
