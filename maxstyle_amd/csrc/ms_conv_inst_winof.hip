// Instantiations: FLAT form of the Winograd F(2x2, 3x3) mode of the wide-read convolution kernel (ms_conv_wide.h, WideGeoWF; storage tags ms_f32wf_t<20 | 24 | 28>; round 6):
// images of 20 / 24 / 28 pixels per row - config 4's deepest levels (512 -> 512 on 20 x 20 at FCN_64 widths, encoder_decoder.py:22-74, 650-653) and the deep levels of the
// reference's shipped 192 / 224-pixel workloads.  The 2x2-output tiles of the whole batch form one list; a work item = 64 consecutive tiles x 32 output channels, so every
// MFMA row is a real tile (the 8-row x 32-pixel tile fills 52 % / 75 % / 77 % of its rows there) and the item count is what the batch holds, not what the tile grid rounds it to
// (20 x 128 -> 128 @28 x 28: 245 items = one round of 256 CUs against 320 = two).  Two channel blocks per staged band, transformed weights from the appendix.  Per output
// element the K loop is the tiled form's: the same bits in `out`.
#include "ms_conv_wide.h"
namespace ms {
int conv_wino_blocks(const ConvArgs& a);      // ms_conv_inst_wino.hip: channel blocks per staged tile the heuristics choose for the TILED form
// Legal: fp32 storage, rows of 20 / 24 / 28 pixels, an even number of rows with at least 64 tiles per image (a work item then spans at most two images), more than 16 output
// channels, weights from the appendix, the plain / accumulate / activation-backward epilogues (the pooled one stays with the tiled form).
static bool flat_legal(const ConvArgs& a) {
  if (a.act_bf16 != 0 || a.wino_nt1 || a.wino_blocks) return false;
  if (!(a.Wout == 20 || a.Wout == 24 || a.Wout == 28) || a.Ws != a.Wout || a.Hout != a.Hs || (a.Hout & 1) || (a.Hout / 2) * (a.Wout / 2) < 64) return false;
  if (a.wu == nullptr || a.Cout <= 16 || !(a.epi_mode == 0 || a.epi_mode == 1 || a.epi_mode == 3)) return false;
  if ((long long)a.N * a.Cout * a.Hout * a.Wout >= (1LL << 31) || 2LL * a.Cin * a.Hs * a.Ws * 4 >= (1LL << 31)) return false;
  return true;
}
// Chosen by ROUNDS of the persistent grid (as the block form, ms_conv_inst_wino.hip).  A flat item is 64 real tiles x 32 channels on one workgroup per CU and costs ~1.15x
// a two-block tile item of the tiled form at 64-128 input channels (its band is 16-20 rows of one image pair against the tile's 10: ~1.35x the staged words; equal at 512
// channels, where the K loop hides the staging).  The tiled form runs ceil(items / CUs) rounds of two-block items, or - where its heuristics take one block - one-block
// workgroups, two per CU: a pair ~1.1x a two-block item, a workgroup alone on its CU ~0.6x.  Measured (tools/ab_wino_nt.py, profiles/r06_wino_ab_*.txt), batch 20:
// 128 -> 128 @28x28 248 flat items = 1 round against 320 = 2 (49.4 -> 30.8 us); 128 -> 64 @24x24 240 one-block workgroups, each alone on a CU, against 90 flat items
// (21.2 vs 31.7 us: tiled); 128 -> 128 @24x24 a tie (30.9 / 30.0 us: tiled).  option "conv.wino_flat": 0 never (A/B runs, same-bits tests), 2 wherever legal.
bool conv_wino_flat(const ConvArgs& a) {
  const int md = opt(OPT_CONV_WINO_FLAT);
  if (md == 0 || !flat_legal(a)) return false;
  if (md >= 2) return true;
  const long cus = num_cus();
  const long items_f = (long)cdiv(a.N * (a.Hout / 2) * (a.Wout / 2), 64) * cdiv(a.Cout, 32);
  const double cost_f = 1.15 * (double)cdiv(items_f, cus);
  const int nt_t = conv_wino_blocks(a);
  const long tiles_t = (long)a.N * cdiv(a.Wout, 32) * cdiv(a.Hout, 8);
  double cost_t;
  if (nt_t == 2) {
    cost_t = (double)cdiv(tiles_t * cdiv(a.Cout, 32), cus);
  } else {
    const long k = cdiv(tiles_t * cdiv(a.Cout, 16), cus);      // one-block workgroups the busiest CU runs
    cost_t = 1.1 * (double)(k / 2) + 0.6 * (double)(k & 1);
  }
  return cost_f < cost_t;
}
template <typename WT>
static int winof(const ConvArgs& a, hipStream_t st) {
  switch (a.pro_mode) {
    case 0: return launch_wino_fx<2, 0, WT>(a, st);
    case 1: return launch_wino_fx<2, 1, WT>(a, st);
    default: return launch_wino_fx<2, 2, WT>(a, st);
  }
}
int conv_dispatch_winof(const ConvArgs& a, hipStream_t st) {
  if (a.Wout == 24) return winof<ms_f32wf_t<24>>(a, st);
  if (a.Wout == 28) return winof<ms_f32wf_t<28>>(a, st);
  return winof<ms_f32wf_t<20>>(a, st);
}
}  // namespace ms
