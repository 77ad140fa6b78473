// C entry point + instantiations of the sub-pixel convolution kernel (ms_conv_subpix.h).
#include "ms_conv_subpix2.h"
#include "maxstyle_hip.h"

using namespace ms;

// 1: every form | 2: an even width that is not a multiple of 4 - the second generation's block geometry only (fp32 storage; mode 0 with the sums appendix) | 0: no
extern "C" int ms_conv_subpix_eligible(int Hs, int Ws) { return (Hs >= 1 && Ws >= 4 && Ws % 4 == 0) ? 1 : ((Hs >= 1 && Ws >= 2 && Ws % 2 == 0) ? 2 : 0); }

// one thread per (product q, input channel, output channel): the sum of the 3x3 taps that land on the same stored pixel, in the first-generation kernel's order
// (ky outer, kx inner, the first tap assigned, the others added in fp32): the kernel that copies the sums from here computes the same bits
#include "ms_pack.h"
namespace ms {
__device__ __forceinline__ void subpix_pack_one(long id, const float* __restrict__ wp, float* __restrict__ sums, int cin_pad, int cout_pad) {
  const long per = (long)cin_pad * cout_pad, total = 16 * per;
  if (id >= total) return;
  const int q = (int)(id / per);
  const long rem = id % per;
  SubCombo cb_{};
#pragma unroll
  for (int qq = 0; qq < 16; ++qq) if (qq == q) cb_ = sub_combo<0>(qq);
  float acc = 0.f;
  bool first = true;
  for (int ky = cb_.ky0; ky <= cb_.ky1; ++ky)
    for (int kx = cb_.kx0; kx <= cb_.kx1; ++kx) {
      const float w = wp[(size_t)(ky * 3 + kx) * per + rem];
      if (first) { acc = w; first = false; } else acc += w;
    }
  sums[id] = acc;
}
__global__ __launch_bounds__(256) void subpix_pack_kernel(const float* __restrict__ wp, float* __restrict__ sums, int cin_pad, int cout_pad) {
  subpix_pack_one((long)blockIdx.x * 256 + threadIdx.x, wp, sums, cin_pad, cout_pad);
}

// Every appendix of a weight version in ONE launch (PackedNets.repack_from_bank issued 2 ms_wino_pack + 1 ms_subpix_pack launches per 3x3 ConvW - 100+ tiny eager
// launches per trainer iteration): a descriptor per job, threads numbered across jobs, binary search for the job as ms_repack_weights does.  Same device functions as the
// single-job entry points: same bits.
struct AppxDesc { long long begin; float* wp; float* sums; int kind; int Cin; int Cout; int cin_pad; int cout_pad; int pad_; };
__global__ __launch_bounds__(256) void appendix_batch_kernel(const AppxDesc* __restrict__ desc, int ndesc, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].begin <= i) lo = mid; else hi = mid - 1; }
  const AppxDesc d = desc[lo];
  const long id = (long)(i - d.begin);
  if (d.kind == 0) wino_pack_one(id, d.wp, d.Cin, d.Cout, d.cin_pad, d.cout_pad);
  else subpix_pack_one(id, d.wp, d.sums, d.cin_pad, d.cout_pad);
}
}  // namespace ms
extern "C" size_t ms_appendix_desc_bytes(void) { return sizeof(AppxDesc); }
// threads a job of this kind needs (the `begin` spacing of the descriptor table): kind 0 = ms_wino_pack, kind 1 = ms_subpix_pack
extern "C" long long ms_appendix_job_threads(int kind, int Cin, int Cout) {
  if (Cin < 1 || Cout < 1) return 0;
  if (kind == 0) return (Cin % 8 != 0) ? 0 : (long long)((Cout + 15) / 16) * (Cin / 8) * 128;
  if (kind == 1) return 16LL * ((Cin + 3) / 4 * 4) * ((Cout + 63) / 64 * 64);
  return 0;
}
extern "C" int ms_appendix_batch(const void* desc_dev, int ndesc, long long total, void* stream) {
  if (ndesc < 1 || total < 1 || desc_dev == nullptr) { set_error("ms_appendix_batch: nothing to do"); return MS_ERR_INVALID; }
  MS_LAUNCH(appendix_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const AppxDesc*)desc_dev, ndesc, total);
  return check_launch("appendix_batch");
}
extern "C" size_t ms_subpix_pack_floats(int Cin, int Cout) {
  if (Cin < 1 || Cout < 1) return 0;
  return (size_t)16 * (size_t)((Cin + 3) / 4 * 4) * (size_t)((Cout + 63) / 64 * 64);
}
extern "C" int ms_subpix_pack(const float* w_packed, float* w_sums, int Cin, int Cout, void* stream) {
  if (w_packed == nullptr || w_sums == nullptr || ms_subpix_pack_floats(Cin, Cout) == 0 || !aligned16(w_sums)) { set_error("ms_subpix_pack: packed 3x3 weights and a 16-byte aligned destination"); return MS_ERR_INVALID; }
  const int cin_pad = (Cin + 3) / 4 * 4, cout_pad = (Cout + 63) / 64 * 64;
  const long total = 16L * cin_pad * cout_pad;
  MS_LAUNCH(subpix_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_packed, w_sums, cin_pad, cout_pad);
  return check_launch("subpix_pack");
}

static int conv_subpix_impl(const float* in, float* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout, int mode,
                            float* stats, const float* ref, const float* u, const float* coef4, float act_slope, float* tab, void* stream, int act_bf16,
                            const float* w_sums = nullptr, int flags = 0) {
  if (N < 1 || Cin < 1 || Cout < 1 || !ms_conv_subpix_eligible(Hs, Ws) || (mode != 0 && mode != 1)) {
    set_error("ms_conv_subpix: invalid shape / mode (stored width must be even)"); return MS_ERR_INVALID;
  }
  if (!aligned16(in) || !aligned16(out) || !aligned16(w_packed) || !aligned16(w_sums)) { set_error("ms_conv_subpix: in, out and the packed weights must be 16-byte aligned"); return MS_ERR_ALIGN; }
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
  a.wu = w_sums;
  hipStream_t st = (hipStream_t)stream;
  // second generation (ms_conv_subpix2.h): fp32 storage, every byte offset of the DMA within 31 bits, and - mode 0 - the sums appendix
  const bool gen2 = !(flags & MS_SUBPIX_FIRST_GEN) && !act_bf16 && (mode == 1 || w_sums != nullptr) &&
                    (long long)N * Cin * Hs * Ws * 4 < (1LL << 31) && 16LL * a.cin_pad * a.cout_pad * 4 < (1LL << 31);
  if (!gen2 && Ws % 4 != 0) { set_error("ms_conv_subpix: a stored width that is not a multiple of 4 needs the second generation (fp32 storage; mode 0: the sums appendix)"); return MS_ERR_INVALID; }
  if (gen2) {
    const int geo = (flags & MS_SUBPIX_TILES) ? 0 : (flags & MS_SUBPIX_BLOCKS) ? 1 : -1;
    return mode == 0 ? launch_conv_subpix2<0>(a, ref, geo, st) : launch_conv_subpix2<1>(a, ref, geo, st);
  }
  return mode == 0 ? launch_conv_subpix<0>(a, ref, st) : launch_conv_subpix<1>(a, ref, st);
}

extern "C" int ms_conv_subpix(const float* in, float* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout, int mode,
                              float* stats, const float* ref, const float* u, const float* coef4, float act_slope, float* tab, void* stream) {
  return conv_subpix_impl(in, out, w_packed, bias, N, Cin, Hs, Ws, Cout, mode, stats, ref, u, coef4, act_slope, tab, stream, 0);
}
extern "C" int ms_conv_subpix2(const float* in, float* out, const float* w_packed, const float* w_sums, const float* bias, int N, int Cin, int Hs, int Ws, int Cout, int mode,
                               float* stats, const float* ref, const float* u, const float* coef4, float act_slope, float* tab, int flags, void* stream) {
  return conv_subpix_impl(in, out, w_packed, bias, N, Cin, Hs, Ws, Cout, mode, stats, ref, u, coef4, act_slope, tab, stream, 0, w_sums, flags);
}
// bf16 activation storage (in, out, ref, u are bf16 bit patterns): see the bf16 section of include/maxstyle_hip.h
extern "C" int ms_conv_subpix_bf16(const uint16_t* in, uint16_t* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout, int mode,
                                   float* stats, const uint16_t* ref, const uint16_t* u, const float* coef4, float act_slope, float* tab, void* stream) {
  return conv_subpix_impl(reinterpret_cast<const float*>(in), reinterpret_cast<float*>(out), w_packed, bias, N, Cin, Hs, Ws, Cout, mode, stats,
                          reinterpret_cast<const float*>(ref), reinterpret_cast<const float*>(u), coef4, act_slope, tab, stream, 1);
}
