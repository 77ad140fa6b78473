// Instantiations: Winograd F(2x2, 3x3) mode of the wide-read convolution kernel (ms_conv_wide.h, storage tags ms_f32w / ms_f32w32 / ms_bf16w / ms_bf16w32).
// (Its own translation unit for build parallelism; the whole library is compiled with -fno-slp-vectorize, see the Makefile.)
#include "ms_conv_wide.h"
namespace ms {
template <typename WT>
static int wide_wino(const ConvArgs& a, hipStream_t st) {
  switch (a.pro_mode) {
    case 0: return launch_wino_fx<1, 0, WT>(a, st);
    case 1: return launch_wino_fx<1, 1, WT>(a, st);
    default: return launch_wino_fx<1, 2, WT>(a, st);
  }
}
int conv_dispatch_wino2(const ConvArgs& a, hipStream_t st);      // ms_conv_inst_wino2.hip: two channel blocks per staged tile
// Channel blocks per staged input tile (round 4).  Two blocks = ONE workgroup per CU (256 registers, 128 accumulators per MFMA wave): the tile is staged, prologue'd and
// transformed once per 32 output channels, but a SIMD then holds one MFMA wave and one staging wave instead of two of each.  Measured per layer on MI355X
// (tools/ab_wino_nt.py, profiles/r04_wino_nt_ab.txt): it wins where the K loop is long and the staging side is light - at least one work item per CU and, with the
// transformed weights staged from the packed tensor's appendix (MS_FETCH_WINO_U: what the engine does), Cin >= 32 under every prologue; when the staging waves transform
// the taps themselves, Cin >= 64, and with the two-tensor BatchNorm-backward prologue only from 256 input channels up or where the layer widens (Cout > Cin).  option "conv.wino_nt": 1 = the one-block form everywhere, 2 = two blocks wherever Cout > 16 (A/B switches); per call: MS_FETCH_WINO_NT1.
bool conv_wino_blockform(const ConvArgs& a);
static int wino_nt(const ConvArgs& a) {
  const int c = opt(OPT_CONV_WINO_NT);
  if (c == 1 || a.Cout <= 16 || a.wino_nt1) return 1;
  if (c >= 2) return 2;
  const int tw = a.Wout < 64 ? 32 : 64, th = 256 / tw;
  long items2 = (long)a.N * cdiv(a.Wout, tw) * cdiv(a.Hout, th) * cdiv(a.Cout, 32);
  if (conv_wino_blockform(a)) items2 = cdiv((long)a.N * cdiv(a.Wout, 8) * cdiv(a.Hout, 8), 4L) * cdiv(a.Cout, 32);
  if (items2 < (long)num_cus()) return 1;
  if (a.wu != nullptr) return a.Cin >= 32 ? 2 : 1;      // weights staged from the appendix (no transform in the staging waves): two blocks win from 32 input channels up, every prologue
  if (a.Cin < 64) return 1;
  if (a.pro_mode == 2 && a.Cin < 256 && a.Cout <= a.Cin) return 1;
  return 2;
}
// Block form (ms_f32wb: four independent 8x8-pixel blocks per work item) where the rectangular tiles waste matrix work: fill = the fraction of a tile grid's pixels that
// exist.  64x4 / 32x8 tiles fill rows of 80, 40 and 20 pixels to 62 %; 8x8 blocks fill any multiple of 8 completely (20 x 20: 69 %).  The block form stages every
// block with its own halo (~17 % more staging work than the 32-pixel tiles), so it is taken only from a clearly better fill (rules below).  option "conv.wino_block": 0 never, 2 wherever legal.
bool conv_wino_blockform(const ConvArgs& a) {
  const int md = opt(OPT_CONV_WINO_BLOCK);
  if (a.wino_blocks) return true;                     // MS_FETCH_WINO_BLOCKS
  if (md == 0 || a.wino_nt1) return false;            // (MS_FETCH_WINO_NT1 pins the round-3 kernel: rectangular tiles, one block)
  if (md >= 2) return true;
  const int tw = a.Wout < 64 ? 32 : 64, th = 256 / tw;
  const double fill_t = (double)a.Wout * a.Hout / ((double)cdiv(a.Wout, tw) * tw * cdiv(a.Hout, th) * th);
  const long nb = (long)a.N * cdiv(a.Wout, 8) * cdiv(a.Hout, 8);
  const double fill_b = (double)a.N * a.Wout * a.Hout / ((double)cdiv(nb, 4) * 4 * 64);
  // measured (tools/ab_wino_nt.py c4, profiles/r04_wino_nt_ab.txt): the blocks win from a 1.15x better fill under the one-tensor prologues (128 -> 128 @160^2: 713 -> 674 us),
  // from 1.5x under the two-tensor prologue (its staging is twice as heavy, and the blocks' private halos add ~20 % to it: 160^2 loses, 80^2 / 40^2 win by 10-25 %);
  // rows of 20 pixels lose either way (three partial blocks per row: 281 vs 251 us)
  if (a.Wout < 32) return false;
  return fill_b >= (a.pro_mode == 2 ? 1.5 : 1.15) * fill_t;
}
int conv_wino_blocks(const ConvArgs& a) { return wino_nt(a); }
int conv_dispatch_winob(const ConvArgs& a, int nt, hipStream_t st);      // ms_conv_inst_winob.hip: the block form
int conv_dispatch_wino(const ConvArgs& a, hipStream_t st) {
  if (conv_wino_blockform(a)) return conv_dispatch_winob(a, wino_nt(a), st);
  if (wino_nt(a) == 2) return conv_dispatch_wino2(a, st);
  if (a.Wout < 64) return a.act_bf16 ? wide_wino<ms_bf16w32>(a, st) : wide_wino<ms_f32w32>(a, st);
  return a.act_bf16 ? wide_wino<ms_bf16w>(a, st) : wide_wino<ms_f32w>(a, st);
}
}  // namespace ms
