// Instantiations: BLOCK form of the Winograd F(2x2, 3x3) mode of the wide-read convolution kernel (ms_conv_wide.h, WideGeoWB; storage tags ms_f32wb / ms_bf16wb):
// a work item = four independent 8x8-pixel blocks (one per MFMA wave, each staged with its own halo by one staging wave) of a flattened (image, block row, block
// column) list, so layers whose rows are not multiples of 32 / 64 pixels (80, 40, 20 at config 4) run without tile-quantisation waste.  One and two channel blocks.
#include "ms_conv_wide.h"
namespace ms {
template <int NT, typename WT>
static int wide_winob(const ConvArgs& a, hipStream_t st) {
  switch (a.pro_mode) {
    case 0: return launch_wino_fx<NT, 0, WT>(a, st);
    case 1: return launch_wino_fx<NT, 1, WT>(a, st);
    default: return launch_wino_fx<NT, 2, WT>(a, st);
  }
}
int conv_dispatch_winob(const ConvArgs& a, int nt, hipStream_t st) {
  if (nt == 2) return a.act_bf16 ? wide_winob<2, ms_bf16wb>(a, st) : wide_winob<2, ms_f32wb>(a, st);
  return a.act_bf16 ? wide_winob<1, ms_bf16wb>(a, st) : wide_winob<1, ms_f32wb>(a, st);
}
}  // namespace ms
