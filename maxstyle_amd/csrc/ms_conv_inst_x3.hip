// Instantiations: three-way bf16 split mode of the wide-read convolution kernel (ms_conv_wide.h, storage tag ms_f32x3).
#include "ms_conv_wide.h"
namespace ms {
int conv_dispatch_x3(const ConvArgs& a, hipStream_t st) {
  switch (a.pro_mode) {
    case 0: return launch_conv_wide_t<1, 0, 1, true, ms_f32x3>(a, st);
    case 1: return launch_conv_wide_t<1, 1, 1, true, ms_f32x3>(a, st);
    default: return launch_conv_wide_t<1, 2, 1, true, ms_f32x3>(a, st);
  }
}
}  // namespace ms
