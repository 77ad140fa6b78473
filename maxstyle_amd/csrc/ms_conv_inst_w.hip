// Instantiations: wide-read 3x3 stride-1 convolution (ms_conv_wide.h), 4-row tiles; the 8-row tiles are in ms_conv_inst_w2.hip.
#include "ms_conv_wide.h"
namespace ms {
// Winograd F(2x2, 3x3) mode of the wide kernel (ms_conv_wide.h, AT = ms_f32w / ms_f32w32): fp32 storage, channel count a multiple of the 8-channel chunk
static bool conv_wino_on(const ConvArgs& a) {
  // option "conv.wino": 0 = direct form everywhere | 1 (default) = where the caller allows it (MS_FETCH_WINOGRAD) | 2 = every eligible call (tools / tests)
  const int mode = opt(OPT_CONV_WINO);
  if (mode == 0 || (mode == 1 && !a.wino_ok)) return false;
  return !(a.act_bf16 == 2 || a.cin_pad % 8 != 0 || a.Cin != a.cin_pad);      // (bf16 matrix arithmetic has its own kernel mode)
}
// rows of 20..63 pixels: only the Winograd form has a tile for them (8 rows x 32 pixels); option "conv.wino32" = 0 leaves them to the first-generation kernel
static bool conv_wino32_on(const ConvArgs& a) {
  const bool on = opt(OPT_CONV_WINO32) != 0;
  // narrower rows leave part of the 32-pixel tile empty: at 20 pixels (62 % full) the form still executes 0.71x the direct form's multiplications and wins
  // (C4's 512-channel 20x20 layers: 27.1 -> 29.3 steps/s), at 16 pixels (0.89x) it loses to the first-generation kernel (C2: 21.9 -> 23.7 us per launch)
  constexpr int minw = 20;
  return on && a.Wout >= minw && a.Wout < 64 && conv_wino_on(a);
}
bool conv_wide_eligible(const ConvArgs& a, int ks, int stride, int fetch, bool vec) {
  if (opt(OPT_CONV_WIDE) == 0) return false;      // A/B switch for timing and for the "same bits as the first generation" tests
  if (ks != 3 || stride != 1 || fetch != FETCH_NORMAL || !vec) return false;
  if (a.epi_mode == 2 || (a.pro_mode != 0 && a.pro_nstride != 0)) return false;
  // (the two-tensor BatchNorm-backward prologue runs with 8-channel chunks: twice the staging registers per channel; 70.4 vs 74.2 us on the
  //  first-generation kernel at 16->16 @16x256x256)
  // fp32 arithmetic: rows of at least one 64-pixel tile.  bf16 matrix arithmetic (act_bf16 == 2): the matrix work of the padding columns of a narrower row is
  // cheap, the first-generation kernel's fp32 MFMAs are not - rows from 16 pixels up take this kernel
  if (a.Wout < (a.act_bf16 == 2 ? 16 : (conv_wino32_on(a) ? 16 : 64)) || a.Wout % 4 != 0) return false;
  if ((long long)a.Cin * a.Hs * a.Ws + a.Ws + 4 >= (1LL << 29)) return false;      // byte offsets inside one image fit 31 bits (buffer addressing of the staging)
  if ((long long)a.Cout * a.Hout * a.Wout >= (1LL << 29)) return false;              // ... and so do the epilogue's offsets inside one output image
  if (!aligned16(a.out)) return false;
  return true;
}
template <int NT>
static int wide_pro(const ConvArgs& a, hipStream_t st) {
  switch (a.pro_mode) {
    case 0: return launch_conv_wide_r<NT, 0, 1>(a, st);
    case 1: return launch_conv_wide_r<NT, 1, 1>(a, st);
    default: return launch_conv_wide_r<NT, 2, 1>(a, st);
  }
}
int conv_dispatch_wino(const ConvArgs& a, hipStream_t st);      // ms_conv_inst_wino.hip
bool conv_wide_is_wino(const ConvArgs& a) { return (a.Wout < 64 && a.act_bf16 != 2) || conv_wino_on(a); }
int conv_dispatch_wide(const ConvArgs& a, int nt, hipStream_t st) {
  if ((a.Wout < 64 && a.act_bf16 != 2) || conv_wino_on(a)) return conv_dispatch_wino(a, st);      // (rows below 64 pixels: conv_wide_eligible admitted them for this form only)
  if (conv_wide_rows(a, nt >= 2 ? 2 : 1) == 8) return conv_dispatch_wide8(a, nt, st);
  return nt >= 2 ? wide_pro<2>(a, st) : wide_pro<1>(a, st);
}
}  // namespace ms
