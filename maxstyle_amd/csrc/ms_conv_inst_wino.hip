// Instantiations: Winograd F(2x2, 3x3) mode of the wide-read convolution kernel (ms_conv_wide.h, storage tags ms_f32w / ms_f32w32 / ms_bf16w / ms_bf16w32).
// (Its own translation unit for build parallelism; the whole library is compiled with -fno-slp-vectorize, see the Makefile.)
#include "ms_conv_wide.h"
namespace ms {
template <typename WT>
static int wide_wino(const ConvArgs& a, hipStream_t st) {
  switch (a.pro_mode) {
    case 0: return launch_conv_wide_t<1, 0, 1, true, WT>(a, st);
    case 1: return launch_conv_wide_t<1, 1, 1, true, WT>(a, st);
    default: return launch_conv_wide_t<1, 2, 1, true, WT>(a, st);
  }
}
int conv_dispatch_wino(const ConvArgs& a, hipStream_t st) {
  if (a.Wout < 64) return a.act_bf16 ? wide_wino<ms_bf16w32>(a, st) : wide_wino<ms_f32w32>(a, st);
  return a.act_bf16 ? wide_wino<ms_bf16w>(a, st) : wide_wino<ms_f32w>(a, st);
}
}  // namespace ms
