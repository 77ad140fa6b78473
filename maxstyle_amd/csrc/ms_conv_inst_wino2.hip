// Instantiations: Winograd F(2x2, 3x3) mode of the wide-read convolution kernel with TWO 16-channel output blocks per staged input tile (NT = 2; ms_conv_wide.h,
// WideGeoW<2, ...>): one 512-thread workgroup per CU on a 256-register budget, 128 accumulators per MFMA wave.  The input tile is staged, prologue'd and
// B^T d B-transformed once per 32 output channels (the NT = 1 form: once per 16).  Per output element the accumulation order is the NT = 1 form's: same bits.
#include "ms_conv_wide.h"
namespace ms {
template <typename WT>
static int wide_wino2(const ConvArgs& a, hipStream_t st) {
  switch (a.pro_mode) {
    case 0: return launch_wino_fx<2, 0, WT>(a, st);
    case 1: return launch_wino_fx<2, 1, WT>(a, st);
    default: return launch_wino_fx<2, 2, WT>(a, st);
  }
}
int conv_dispatch_wino2(const ConvArgs& a, hipStream_t st) {
  if (a.Wout < 64) return a.act_bf16 ? wide_wino2<ms_bf16w32>(a, st) : wide_wino2<ms_f32w32>(a, st);
  return a.act_bf16 ? wide_wino2<ms_bf16w>(a, st) : wide_wino2<ms_f32w>(a, st);
}
}  // namespace ms
