// Instantiations: FLAT form of the Winograd F(2x2, 3x3) mode of the wide-read convolution kernel (ms_conv_wide.h, WideGeoWF; storage tag ms_f32wf; round 6): images of
// 20 x 20 pixels (config 4's deepest levels - 512 -> 512 at FCN_64 widths, encoder_decoder.py:22-74, 650-653).  The 2x2-output tiles of the whole batch form one list; a
// work item = 64 consecutive tiles x 32 output channels, so every MFMA row is a real tile (the 8-row x 32-pixel tile fills 52 % of its rows at 20 x 20).  Two channel
// blocks per staged band, transformed weights from the appendix.  Per output element the K loop is the tiled form's: the same bits in `out`.
#include "ms_conv_wide.h"
namespace ms {
int conv_wino_blocks(const ConvArgs& a);      // ms_conv_inst_wino.hip: channel blocks per staged tile the heuristics choose
// Eligible: fp32 storage, 20-pixel rows, an even number of rows >= 14 (64 tiles span at most two images), two channel blocks, weights from the appendix, the plain /
// accumulate / activation-backward epilogues (the pooled one stays with the tiled form); option "conv.wino_flat" = 0 switches it off (A/B runs, same-bits tests)
bool conv_wino_flat(const ConvArgs& a) {
  if (opt(OPT_CONV_WINO_FLAT) == 0 || a.act_bf16 != 0 || a.wino_nt1 || a.wino_blocks) return false;
  if (a.Wout != 20 || a.Ws != 20 || a.Hout != a.Hs || (a.Hout & 1) || a.Hout < 14) return false;
  if (a.wu == nullptr || !(a.epi_mode == 0 || a.epi_mode == 1 || a.epi_mode == 3)) return false;
  if ((long long)a.N * a.Cout * a.Hout * a.Wout >= (1LL << 31) || 2LL * a.Cin * a.Hs * a.Ws * 4 >= (1LL << 31)) return false;
  return conv_wino_blocks(a) == 2;
}
int conv_dispatch_winof(const ConvArgs& a, hipStream_t st) {
  switch (a.pro_mode) {
    case 0: return launch_wino_fx<2, 0, ms_f32wf>(a, st);
    case 1: return launch_wino_fx<2, 1, ms_f32wf>(a, st);
    default: return launch_wino_fx<2, 2, ms_f32wf>(a, st);
  }
}
}  // namespace ms
