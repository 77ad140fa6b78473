// C entry point + instantiations of the sub-pixel convolution kernel (ms_conv_subpix.h).
#include "ms_conv_subpix.h"
#include "maxstyle_hip.h"

using namespace ms;

extern "C" int ms_conv_subpix_eligible(int Hs, int Ws) { return (Hs >= 1 && Ws >= 4 && Ws % 4 == 0) ? 1 : 0; }

static int conv_subpix_impl(const float* in, float* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout, int mode,
                            float* stats, const float* ref, const float* u, const float* coef4, float act_slope, float* tab, void* stream, int act_bf16) {
  if (N < 1 || Cin < 1 || Cout < 1 || !ms_conv_subpix_eligible(Hs, Ws) || (mode != 0 && mode != 1)) {
    set_error("ms_conv_subpix: invalid shape / mode (stored width must be a multiple of 4)"); return MS_ERR_INVALID;
  }
  if (!aligned16(in) || !aligned16(out) || !aligned16(w_packed)) { set_error("ms_conv_subpix: in, out and the packed weights must be 16-byte aligned"); return MS_ERR_ALIGN; }
  const bool mask = (ref != nullptr) || (u != nullptr);        // ref == NULL with u given: the mask is recomputed from sc*u + sh (activation never materialised)
  if (mask && (u == nullptr || coef4 == nullptr || tab == nullptr || stats != nullptr || (ref != nullptr && !aligned16(ref)) || !aligned16(u) || !aligned16(coef4))) {
    set_error("ms_conv_subpix: the activation-backward epilogue needs ref, u, coef4 and tab (16-byte aligned) and no statistics"); return MS_ERR_INVALID;
  }
  if ((long long)Cin * Hs * Ws >= (1LL << 31) || (long long)Cout * 4 * Hs * Ws >= (1LL << 31)) { set_error("ms_conv_subpix: plane offsets exceed 31 bits"); return MS_ERR_INVALID; }
  ConvArgs a{};
  a.in = in; a.out = out; a.w = w_packed; a.bias = bias; a.stats = stats;
  a.N = N; a.Cin = Cin; a.Hs = Hs; a.Ws = Ws; a.Hin = Hs; a.Win = Ws; a.Cout = Cout; a.Hout = 2 * Hs; a.Wout = 2 * Ws;
  a.cin_pad = (Cin + 3) / 4 * 4; a.cout_pad = (Cout + 63) / 64 * 64; a.cout_real = Cout;
  a.epi_mode = mask ? 3 : 0;
  a.mk_u = u; a.mk_coef = coef4; a.mk_slope = act_slope; a.mk_tab = tab;
  a.act_bf16 = act_bf16;
  hipStream_t st = (hipStream_t)stream;
  return mode == 0 ? launch_conv_subpix<0>(a, ref, st) : launch_conv_subpix<1>(a, ref, st);
}

extern "C" int ms_conv_subpix(const float* in, float* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout, int mode,
                              float* stats, const float* ref, const float* u, const float* coef4, float act_slope, float* tab, void* stream) {
  return conv_subpix_impl(in, out, w_packed, bias, N, Cin, Hs, Ws, Cout, mode, stats, ref, u, coef4, act_slope, tab, stream, 0);
}
// bf16 activation storage (in, out, ref, u are bf16 bit patterns): see the bf16 section of include/maxstyle_hip.h
extern "C" int ms_conv_subpix_bf16(const uint16_t* in, uint16_t* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout, int mode,
                                   float* stats, const uint16_t* ref, const uint16_t* u, const float* coef4, float act_slope, float* tab, void* stream) {
  return conv_subpix_impl(reinterpret_cast<const float*>(in), reinterpret_cast<float*>(out), w_packed, bias, N, Cin, Hs, Ws, Cout, mode, stats,
                          reinterpret_cast<const float*>(ref), reinterpret_cast<const float*>(u), coef4, act_slope, tab, stream, 1);
}
