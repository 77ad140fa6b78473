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
static int wino_nt_for(const ConvArgs& a, bool blocks) {      // wino_nt's rule for a given form (no recursion into the form choice)
  const int c = opt(OPT_CONV_WINO_NT);
  if (c == 1 || a.Cout <= 16 || a.wino_nt1) return 1;
  if (c >= 2) return 2;
  const int tw = a.Wout < 64 ? 32 : 64, th = 256 / tw;
  long items2 = (long)a.N * cdiv(a.Wout, tw) * cdiv(a.Hout, th) * cdiv(a.Cout, 32);
  if (blocks) items2 = cdiv((long)a.N * cdiv(a.Wout, 8) * cdiv(a.Hout, 8), 4L) * cdiv(a.Cout, 32);
  if (items2 < (long)num_cus()) return 1;
  if (a.wu != nullptr) return a.Cin >= 32 ? 2 : 1;
  if (a.Cin < 64) return 1;
  if (a.pro_mode == 2 && a.Cin < 256 && a.Cout <= a.Cin) return 1;
  return 2;
}
bool conv_wino_blockform(const ConvArgs& a) {
  const int md = opt(OPT_CONV_WINO_BLOCK);
  if (a.wino_blocks) return true;                     // MS_FETCH_WINO_BLOCKS
  if (md == 0 || a.wino_nt1) return false;            // (MS_FETCH_WINO_NT1 pins the round-3 kernel: rectangular tiles, one block)
  if (md >= 2) return true;
  if (a.Wout < 32) return false;                      // rows of 20 / 24 / 28 pixels lose either way (three partial blocks per row: 281 vs 251 us at 20 x 20)
  // Round 6: the choice by ROUNDS of the persistent grid, not by fill alone.  A launch runs ceil(items / resident workgroups) rounds of work items; an item of the
  // block form costs ~1.15x a tile item under the one-tensor prologues and ~1.3x under the two-tensor prologue (private halos: 2.5 scalar loads per lane and chunk
  // instead of 0.6).  The fill rule of round 4 (blocks from a 1.15x / 1.5x better fill) is this rule on long launches, where rounds ~ items; on the reference's SHIPPED
  // shapes (batch 20, 1-5 rounds) the integer matters: 64 -> 64 @20x56x56 is 560 tile items = 3 rounds against 490 block items = 2 (41.3 -> 32.5 us), while
  // 64 -> 64 @20x48x48 is 2 rounds either way and the blocks' better fill loses (29.1 vs 32.4 us) - tools/ab_wino_nt.py acdc192 / prostate224, profiles/r06_wino_ab_*.txt.
  const int tw = a.Wout < 64 ? 32 : 64, th = 256 / tw;
  const int nt_t = wino_nt_for(a, false), nt_b = wino_nt_for(a, true);
  const long items_t = (long)a.N * cdiv(a.Wout, tw) * cdiv(a.Hout, th) * cdiv(a.Cout, 16 * nt_t);
  const long items_b = cdiv((long)a.N * cdiv(a.Wout, 8) * cdiv(a.Hout, 8), 4L) * cdiv(a.Cout, 16 * nt_b);
  const long slots_t = (long)num_cus() * (nt_t == 1 ? 2 : 1), slots_b = (long)num_cus() * (nt_b == 1 ? 2 : 1);
  // (a round = every CU works through 32 output channels of one tile: one two-block workgroup, or two one-block workgroups sharing the CU - ~1.1x as long)
  const double cost_t = (double)cdiv(items_t, slots_t) * (nt_t == 1 ? 1.1 : 1.0);
  const double cost_b = (double)cdiv(items_b, slots_b) * (nt_b == 1 ? 1.1 : 1.0) * (a.pro_mode == 2 ? 1.3 : 1.15);
  return cost_b < cost_t;
}
int conv_wino_blocks(const ConvArgs& a) { return wino_nt(a); }
int conv_dispatch_winob(const ConvArgs& a, int nt, hipStream_t st);      // ms_conv_inst_winob.hip: the block form
bool conv_wino_flat(const ConvArgs& a);                                  // ms_conv_inst_winof.hip: the flat form (20-pixel rows)
int conv_dispatch_winof(const ConvArgs& a, hipStream_t st);
int conv_dispatch_wino(const ConvArgs& a, hipStream_t st) {
  if (conv_wino_flat(a)) return conv_dispatch_winof(a, st);
  if (conv_wino_blockform(a)) return conv_dispatch_winob(a, wino_nt(a), st);
  if (wino_nt(a) == 2) return conv_dispatch_wino2(a, st);
  if (a.Wout < 64) return a.act_bf16 ? wide_wino<ms_bf16w32>(a, st) : wide_wino<ms_f32w32>(a, st);
  return a.act_bf16 ? wide_wino<ms_bf16w>(a, st) : wide_wino<ms_f32w>(a, st);
}
}  // namespace ms
