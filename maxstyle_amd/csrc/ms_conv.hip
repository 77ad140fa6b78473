// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Replaces (reference, eager ATen): nn.Conv2d k=3/p=1 s=1|2, k=1 (encoder_decoder.py:40-47, 323-337, 441-445, 650-653),
// nn.ConvTranspose2d k=2 s=2 (:306) and - with transformed weights - their data-gradients (convolution_backward).
// No weight-gradient: inside the inner loop every network parameter has requires_grad=False
// (advanced_triplet_recon_segmentation_model.py:508-511).
//
// One kernel template covers all of them:
//   out[n,co,oy,ox] = bias[co] + sum_{ci,ky,kx} P(in)[n,ci, oy*S+ky-PAD, ox*S+kx-PAD] * w[ky,kx,ci,co]
//   FETCH   how the logical input is read from the stored tensor:
//           NORMAL; UPS2 = nearest x2 up-sampling fused into the load (nn.UpsamplingNearest2d, :298-300);
//           ZINS2 = zero-insertion x2 (turns the stride-2 conv's data-gradient into a stride-1 conv).
//   P       prologue fused into LDS staging: none | LeakyReLU(a[c]*v+b[c]) (BatchNorm apply + activation of the
//           producer layer, per channel or per (n,c) plane) | a[c]*v + b[c]*in2 + d[c] (BatchNorm backward apply).
//   epilogue: +bias, store | accumulate into out | 2x2 pixel-shuffle store (ConvTranspose2d k2 s2), plus optional
//           per-block (count, mean, M2) of the outputs per channel for the BatchNorm batch statistics
//           (two-pass in registers, Chan-merged later in fp64: no E[x^2]-E[x]^2 cancellation, no float atomics).
//
// Tiling: a workgroup (4 waves) computes an 8x32 output-pixel tile for 16*NT output channels; wave w owns output rows
// 2w,2w+1 = four 16-pixel M-tiles; K runs over (tap, 4 input channels) per MFMA. The input halo tile and the weight
// slice of CK input channels are staged in LDS; A fragments are ds_read_b32 of 16 consecutive pixels per k (plane
// stride == 16 mod 32 banks -> conflict-free), B fragments 16 consecutive output channels per k.
#include <algorithm>
#include <cstdlib>
#include "ms_conv_kernel.h"
#include "ms_conv_wide.h"
#include "ms_conv_k1s.h"
#include "ms_conv_s2.h"
#include "ms_conv_k1g.h"
#include "ms_conv_k3n.h"
#include "ms_pack.h"
#include "maxstyle_hip.h"

namespace ms {

// ---------------------------------------------------------------------------------------------------------
// BatchNorm statistics finalize: Chan-merge the per-block (count, mean, M2) in fp64 -> per channel
//   out[c] = {scale = gamma*invstd, shift = beta - mean*scale, mean, invstd}      (biased variance, eps)
// model_util.py:468-510 mode: batch statistics, running buffers untouched.
// ---------------------------------------------------------------------------------------------------------
// One WAVE per channel, one pass: the table holds one slot per workgroup of the conv (<= 512), so 64 lanes x 8 slots with shuffle reductions beat
// a 256-thread block with two passes and three LDS reductions (this kernel sits on the ~4 us latency floor of a dependent launch, 23 times per step).
// M2 = sum M2_i + sum n_i m_i^2 - (sum n_i m_i)^2 / N in fp64: the inputs are fp32, so the cancellation costs (mean/sigma)^2 * 1e-16.
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float4* __restrict__ tab, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float4* __restrict__ out) {
  const int c = blockIdx.x;
  const int nparts = (int)tab[0].x;
  const float4* part = tab + 1 + (size_t)c * kStatSlots;
  double sn = 0.0, sm = 0.0, sq = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) {
    const float4 q = part[i];
    const double n = (double)q.x, mu = (double)q.y;
    sn += n; sm += n * mu; sq += (double)q.z + n * mu * mu;
  }
  sn = wave_sum_d(sn); sm = wave_sum_d(sm); sq = wave_sum_d(sq);
  if (threadIdx.x == 0) {
    const double mean = sm / sn;
    const double var = fmax((sq - sm * mean) / sn, 0.0);
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    out[c] = make_float4(sc, beta[c] - (float)mean * sc, (float)mean, invstd);
  }
}

// tile geometry is a pure function of the output width (so ms_conv_stats_parts can be answered without launching)
static inline bool narrow_tile(int Wout) { return Wout <= 16; }
static inline int tile_h(int Wout) { return narrow_tile(Wout) ? 4 : 8; }
static inline int tile_w(int Wout) { return narrow_tile(Wout) ? 16 : 32; }

}  // namespace ms

using namespace ms;

// capacity (slots per channel) of the partial table: one slot per consumer wave of a resident workgroup, whatever the shape
extern "C" int ms_conv_stats_parts(int N, int Hout, int Wout) { (void)N; (void)Hout; (void)Wout; return kStatSlots; }

extern "C" size_t ms_conv_stats_bytes(int N, int Cout, int Hout, int Wout) {
  return ((size_t)Cout * ms_conv_stats_parts(N, Hout, Wout) + 1) * sizeof(float4);      // + header record
}

struct MaskEpi { const float* u; const float* coef4; float slope; float* tab; int mode = 3; };     // mode 3: activation-backward epilogue; 4 / 5: residual-block tail
// cross-workgroup finalize (ConvArgs::xf_*): the statistics table of the BatchNorm whose coefficients this launch consumes, its affine parameters, the record
// buffer the launch fills for later kernels, the granule table (2 x 8 bytes per channel, zero-filled once by the caller) and the error word
struct Ride { int kind; const float* part2; int nparts; const float* coef4; const float* p1; float eps; double count; float* out4; int C; };      // ConvArgs::ride_*
struct XFin { const float* tab; const float* gamma; const float* beta; float eps; float* coef4; void* gran; int* err; int C; int kind = 0; double count = 0.0; };

static int conv2d_impl(const float* in, const float* in2, float* out, const float* w_packed, const float* bias,
                       int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                       int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_nstride, int pro_cstride, float slope,
                       int epi_mode, float* stats, const MaskEpi* mk, void* stream, int act_bf16 = 0, const XFin* xf = nullptr, const Ride* ride = nullptr) {
  if (N < 1 || Cin < 1 || Cout < 1 || Hs < 1 || Ws < 1) { set_error("ms_conv2d: invalid shape"); return MS_ERR_INVALID; }
  // bit 8 of `fetch` (MS_FETCH_WINOGRAD): the caller accepts the Winograd F(2x2,3x3) form for this call where it is built (see include/maxstyle_hip.h)
  const bool wino_ok = (fetch >= 0) && (fetch & MS_FETCH_WINOGRAD) != 0;
  const bool wino_nt1 = (fetch >= 0) && (fetch & MS_FETCH_WINO_NT1) != 0;      // bit 10: the one-channel-block variant of the Winograd form
  const bool wino_u = (fetch >= 0) && (fetch & MS_FETCH_WINO_U) != 0;          // bit 11: w_packed carries the Winograd appendix
  const bool wino_blocks = (fetch >= 0) && (fetch & MS_FETCH_WINO_BLOCKS) != 0;     // bit 12: the block form wherever legal
  if (fetch >= 0) fetch &= ~(MS_FETCH_WINOGRAD | MS_FETCH_WINO_NT1 | MS_FETCH_WINO_U | MS_FETCH_WINO_BLOCKS);
  if (pro_mode < 0 || pro_mode > 2 || epi_mode < 0 || (epi_mode > 2 && epi_mode != MS_EPI_POOL2) || fetch < 0 || fetch > 2) { set_error("ms_conv2d: invalid mode"); return MS_ERR_INVALID; }
  if (epi_mode == MS_EPI_POOL2 && (mk != nullptr || stats != nullptr || bias != nullptr)) { set_error("ms_conv2d: the pooled epilogue is a plain store (no bias, statistics or mask)"); return MS_ERR_INVALID; }
  // the activation helper computes max(v, v*slope): LeakyReLU / ReLU slopes only
  if ((pro_mode == 1 && !(slope >= 0.f && slope <= 1.f)) || (mk != nullptr && !(mk->slope >= 0.f && mk->slope <= 1.f))) { set_error("ms_conv2d: activation slope outside [0, 1]"); return MS_ERR_INVALID; }
  if (mk != nullptr && mk->mode != 3) {
    if (epi_mode != 0 || stats != nullptr || ks != 1 || stride != 1 || fetch != 0 || pro_mode != 0 || mk->u == nullptr || mk->coef4 == nullptr ||
        !aligned16(mk->u) || !aligned16(mk->coef4) || !aligned16(out)) {
      set_error("ms_conv1x1_bnres: 1x1 stride-1 conv without prologue; u, coef4 and out 16-byte aligned"); return MS_ERR_INVALID;
    }
    epi_mode = mk->mode;
  } else if (mk != nullptr) {
    if (epi_mode != 0 || stats != nullptr || mk->u == nullptr || mk->coef4 == nullptr || mk->tab == nullptr || !aligned16(mk->u) || !aligned16(mk->coef4) || !aligned16(out)) {
      set_error("ms_conv2d_actbwd: needs u, coef4 and tab (16-byte aligned) and a plain epilogue"); return MS_ERR_INVALID;
    }
    epi_mode = 3;
  }
  const bool xf_pro = (xf != nullptr) && (mk == nullptr || mk->mode == 3);        // the `_xfin` launch feeds its PROLOGUE (not a residual tail's epilogue)
  if (xf_pro) {
    if (!((xf->kind == 0 && pro_mode == 1) || (xf->kind == 1 && pro_mode == 2 && xf->count > 0)) || pro_nstride != 0 || act_bf16 == 2) {
      set_error("ms_conv2d_xfin: kind 0 with pro_mode 1 or kind 1 with pro_mode 2 (count > 0), per-channel coefficients, not the bf16-MFMA mode"); return MS_ERR_INVALID;
    }
    pro_a = pro_b = pro_c = xf->coef4;                                             // (not read by the launch: the table is filled from the granules)
    pro_cstride = 4;
  }
  if (pro_mode == 2 && (in2 == nullptr || pro_c == nullptr)) { set_error("ms_conv2d: pro_mode 2 needs in2 and pro_c"); return MS_ERR_INVALID; }
  if (pro_mode != 0 && (pro_a == nullptr || pro_b == nullptr)) { set_error("ms_conv2d: prologue coefficients missing"); return MS_ERR_INVALID; }
  if (epi_mode == 2 && (ks != 1 || stride != 1 || fetch != 0 || stats != nullptr)) { set_error("ms_conv2d: pixel-shuffle epilogue is for the k=1 GEMM form of ConvTranspose2d(k=2,s=2)"); return MS_ERR_INVALID; }
  if (!aligned16(w_packed)) { set_error("ms_conv2d: packed weights must be 16-byte aligned"); return MS_ERR_ALIGN; }
  const bool shape_ok = (ks == 3 && (stride == 1 || stride == 2)) || (ks == 1 && stride == 1) || (ks == 2 && stride == 2);
  if (!shape_ok || (fetch != FETCH_NORMAL && !(ks == 3 && stride == 1)) || (pro_mode == 2 && stride != 1)) {
    set_error("ms_conv2d: unsupported (ks=%d, stride=%d, fetch=%d, pro_mode=%d)", ks, stride, fetch, pro_mode);
    return MS_ERR_INVALID;
  }
  ConvArgs a{};
  a.in = in; a.in2 = (pro_mode == 2) ? in2 : nullptr; a.out = out; a.w = w_packed; a.bias = bias;
  a.pro_a = pro_a; a.pro_b = pro_b; a.pro_c = pro_c; a.stats = stats;
  a.N = N; a.Cin = Cin; a.Hs = Hs; a.Ws = Ws;
  a.Hin = (fetch == FETCH_NORMAL) ? Hs : 2 * Hs; a.Win = (fetch == FETCH_NORMAL) ? Ws : 2 * Ws;
  const int pad = (ks == 3) ? 1 : 0;
  a.Hout = (a.Hin + 2 * pad - ks) / stride + 1; a.Wout = (a.Win + 2 * pad - ks) / stride + 1;
  a.cout_real = Cout;
  const int gemm_cols = (epi_mode == 2) ? 4 * Cout : Cout;
  a.Cout = gemm_cols;
  a.cin_pad = (Cin + 3) / 4 * 4; a.cout_pad = (gemm_cols + 63) / 64 * 64;
  a.wino_ok = wino_ok ? 1 : 0;
  a.wino_nt1 = wino_nt1 ? 1 : 0;
  a.wino_blocks = wino_blocks ? 1 : 0;
  a.wu = (wino_u && wino_ok && ks == 3 && stride == 1 && Cin % 8 == 0) ? w_packed + (size_t)9 * a.cin_pad * a.cout_pad : nullptr;
  a.act_bf16 = act_bf16;                  // 0 fp32 storage | 1 bf16 storage, fp32 matrix arithmetic | 2 bf16 storage, bf16 matrix arithmetic where built (`_bf16m`)
  a.pro_mode = pro_mode; a.pro_nstride = pro_nstride; a.pro_cstride = pro_cstride < 1 ? 1 : pro_cstride; a.slope = slope; a.epi_mode = epi_mode;
  if (mk != nullptr) { a.mk_u = mk->u; a.mk_coef = mk->coef4; a.mk_slope = mk->slope; a.mk_tab = mk->tab; }
  if (xf != nullptr) {
    if (xf->tab == nullptr || xf->gamma == nullptr || (xf->kind == 0 && xf->beta == nullptr) || xf->coef4 == nullptr || xf->gran == nullptr || xf->err == nullptr || xf->C < 1 ||
        !aligned16(xf->tab) || !aligned16(xf->coef4) || (reinterpret_cast<uintptr_t>(xf->gran) & 7u) != 0) {
      set_error("ms_conv*_xfin: statistics table, gamma, beta, coef4 (16-byte aligned), granule table (8-byte aligned) and the error word are required"); return MS_ERR_INVALID;
    }
    a.xf_tab = xf->tab; a.xf_gamma = xf->gamma; a.xf_beta = xf->beta; a.xf_eps = xf->eps; a.xf_coef = xf->coef4;
    a.xf_gran = reinterpret_cast<conv_u64_t*>(xf->gran); a.xf_err = xf->err; a.xf_C = xf->C; a.xf_kind = xf->kind; a.xf_count = xf->count;
  }
  if (ride != nullptr) {
    const bool k0 = ride->kind == 0 && ride->nparts >= 0 && ride->count > 0 && aligned16(ride->coef4) && (reinterpret_cast<uintptr_t>(ride->part2) & 7u) == 0;
    const bool k1 = ride->kind == 1 && ride->p1 != nullptr && aligned16(ride->part2);
    if (ks != 1 || ride->part2 == nullptr || ride->coef4 == nullptr || ride->out4 == nullptr || ride->C < 1 || !aligned16(ride->out4) || !(k0 || k1)) {
      set_error("ms_conv2d_ride: a 1x1 conv; kind 0: partial sums (8-byte aligned), forward records and out4 (16-byte aligned), count > 0; kind 1: statistics table, gamma, beta"); return MS_ERR_INVALID;
    }
    a.ride_kind = ride->kind; a.ride_beta = ride->p1; a.ride_eps = ride->eps;
    a.ride_part = reinterpret_cast<const float2*>(ride->part2); a.ride_nparts = ride->nparts; a.ride_coef = reinterpret_cast<const float4*>(ride->coef4);
    a.ride_count = ride->count; a.ride_out = reinterpret_cast<float4*>(ride->out4); a.ride_C = ride->C;
  }
  if (a.Hout < 1 || a.Wout < 1) { set_error("ms_conv2d: empty output"); return MS_ERR_INVALID; }
  if ((long)N > 65535) { set_error("ms_conv2d: batch too large for gridDim.z"); return MS_ERR_INVALID; }
  const bool narrow = narrow_tile(a.Wout);
  a.tiles_x = cdiv(a.Wout, tile_w(a.Wout)); a.tiles_y = cdiv(a.Hout, tile_h(a.Wout));
  // (16-byte staging loads use buffer addressing with 31-bit byte offsets inside one image: ms_conv_kernel.h BUF_LD)
  const bool vec = (fetch == FETCH_NORMAL) && (Ws % 4 == 0) && aligned16(in) && (a.in2 == nullptr || aligned16(a.in2)) && ((long long)Cin * Hs * Ws < (1LL << 29) - 64);
  // output-channel tile: the widest (best reuse of the staged input tile) that still leaves >= 2 work items per CU;
  // failing that, the widest that leaves >= 1 per CU; else 16 channels (most parallelism)
  const long tiles = (long)a.tiles_x * a.tiles_y * N;
  int nt = 1;
  bool found = false;
  for (long want : {512L, 256L}) {
    // 64-channel tiles only where they fit the register file without spilling; not for the pixel-shuffle epilogue (its scattered 4-byte stores
    // want more workgroups in flight: 16->4x16 @16x128x128 56.6 us with 64-column tiles, 46.5 us with 32; tools/tune_conv.py)
    for (int cand = ((pro_mode == 2 || ks == 3 || epi_mode == 2) ? 2 : 4); cand >= 2 && !found; cand >>= 1) {
      if (gemm_cols <= 16 * (cand / 2)) continue;        // would be mostly padding
      if (tiles * cdiv(gemm_cols, 16 * cand) >= want) { nt = cand; found = true; }
    }
    // two resident workgroups per CU with 16-channel tiles beat one with 32-channel tiles (285.4 -> 287.3 steps/s)
    // (not for a stride-2 layer of <= 32 channels: its input tile is 4x the output tile, and one 32-channel block stages it once instead of
    //  twice - 32->32 @16x128x128: 20.3 vs 27.1 us; with more channel blocks the extra workgroups win again: 128->128 @16x32x32 23.6 vs 26.7 us)
    // (since the staging waves stopped being the bound - buffer addressing, round 2 - a deep 3x3 layer prefers ONE 32-channel workgroup per CU, which stages
    //  each input tile once instead of twice: 128->128 @16x32x32 data-gradient + activation backward 60.4 -> 53.8 us, 64->128 @16x32x32 32.3 -> 28.7; not the 16-pixel-wide layers (25.8 vs 22.6)
    //  nor prologue-free convs (28.4 vs 27.1); tools/tune_conv.py)
    if (!found && want == 512L && ks == 3 && stride == 1 && Cin >= 64 && gemm_cols > 16 && pro_mode != 0 && !narrow && tiles * cdiv(gemm_cols, 32) >= 256L) { nt = 2; found = true; }
    if (!found && want == 512L && !(stride == 2 && gemm_cols <= 32) && tiles * cdiv(gemm_cols, 16) >= 512L) { nt = 1; found = true; }
    if (found) break;
  }
  // 1x1 convolutions on small images with many channels (config 4's 20x20 / 40x40 levels, 256-512 channels: fewer tiles than CUs): 16-channel tiles, i.e. the most
  // work items - the 64-channel tile needs 205 registers (one workgroup per CU) and leaves most of the chip idle there (tools/tune_conv.py 10 c4: 120 -> 92 us at
  // 512 -> 512 @16x20x20, 201 -> 150 us at 512 -> 256 @16x40x40).  Not the 16-pixel-wide levels (their own tile shape and chunking, tuned in round 3).
  if (ks == 1 && !narrow && epi_mode != 2 && tiles < (long)num_cus() && gemm_cols >= 256 && tiles * cdiv(gemm_cols, 16) >= 512L) nt = 1;
  // tuning hook (tools/tune_conv.py): option "conv.force_nt"
  {
    const int fnt = opt(OPT_CONV_FORCE_NT);
    if (fnt == 1 || fnt == 2 || (fnt == 4 && ks != 3 && pro_mode != 2)) nt = fnt;
  }
  constexpr bool allow_wide = true;      // (every second-generation form has its own option: conv.wide / conv.k1s / conv.k1g / conv.s2g2)
  a.ncb = cdiv(gemm_cols, 16 * nt);
  a.dbg = opt(OPT_DIAG_CONV_DBG);
  a.trace = conv_trace_buffer();
  hipStream_t st = (hipStream_t)stream;
  const bool use_in2 = (pro_mode == 2);
  if (epi_mode == MS_EPI_POOL2) {
    if (!(allow_wide && conv_wide_eligible(a, ks, stride, fetch, vec) && conv_wide_is_wino(a) && a.Hout % 2 == 0)) {
      set_error("ms_conv2d: the pooled epilogue needs the Winograd form of the wide kernel (ms_conv2d_pool2_ok)"); return MS_ERR_INVALID;
    }
    return conv_dispatch_wide(a, nt, st);
  }
  if (allow_wide && conv_wide_eligible(a, ks, stride, fetch, vec)) return conv_dispatch_wide(a, nt, st);
  if (conv_k3n_eligible(a, ks, stride, fetch)) return conv_dispatch_k3n(a, ks, st, stride);      // second generation for rows of 12 / 14 / 16 pixels (ms_conv_k3n.h)
  if (ks == 3 && stride == 1) return conv_dispatch_k3s1(a, fetch, nt, vec, narrow, use_in2, st);
  if (ks == 1 && allow_wide && conv_k1s_eligible(a, ks, stride, fetch)) return conv_dispatch_k1s(a, st);      // streaming form (ms_conv_k1s.h)
  if (ks == 1 && allow_wide && conv_k1g_eligible(a, ks, stride, fetch)) return conv_dispatch_k1g(a, st);      // LDS-tiled GEMM form of the channel-heavy levels (ms_conv_k1g.h)
  if (ks == 1) return conv_dispatch_k1s1(a, nt, vec, narrow, use_in2, st);
  if (allow_wide && conv_s2g2_eligible(a, ks, stride, fetch)) return conv_dispatch_s2g2(a, st);      // second generation (ms_conv_s2.h)
  return conv_dispatch_s2(a, ks, nt, vec, narrow, st);
}

namespace ms { int conv_k1s_switch() { return opt(OPT_CONV_K1S); } }
// whether ms_conv2d (epi_mode 0 / 2) / ms_conv1x1_bnres (epi_mode 4; 5 = half-resolution input) would take the streaming kernel for this 1x1 shape (16-byte aligned fp32 tensors assumed)
extern "C" int ms_conv_k1s_would_run(int N, int Cin, int H, int W, int Cout, int epi_mode) {
  if (N < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1) return 0;
  ConvArgs a{};
  a.N = N; a.Cin = Cin; a.Hs = a.Hin = a.Hout = H; a.Ws = a.Win = a.Wout = W; a.cout_real = Cout; a.Cout = (epi_mode == 2) ? 4 * Cout : Cout;
  a.cin_pad = (Cin + 3) / 4 * 4; a.cout_pad = (a.Cout + 63) / 64 * 64; a.epi_mode = epi_mode;
  return conv_k1s_eligible(a, 1, 1, FETCH_NORMAL) ? 1 : 0;
}
namespace ms { int conv_s2g2_switch() { return opt(OPT_CONV_S2G2); } }
namespace ms { int conv_k1g_switch() { return opt(OPT_CONV_K1G); } }

extern "C" int ms_conv2d(const float* in, const float* in2, float* out, const float* w_packed, const float* bias,
                         int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                         int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_nstride, int pro_cstride, float slope,
                         int epi_mode, float* stats, void* stream) {
  return conv2d_impl(in, in2, out, w_packed, bias, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pro_mode, pro_a, pro_b, pro_c, pro_nstride, pro_cstride, slope,
                     epi_mode, stats, nullptr, stream);
}

// ms_conv2d (a 1x1 conv) that also carries a ms_bn_bwd_coefs (ride_kind 0) or ms_bn_finalize (ride_kind 1) job for the launch BEHIND it: one MFMA wave per channel
// derives out4[c] while the staging waves fetch (ConvArgs::ride_*).  The conv itself neither reads nor waits for out4; the channels are dealt round-robin over the launch's MFMA waves (any grid).
// whether ms_conv2d(ks 3, stride 1, fetch MS_FETCH_WINOGRAD, epi_mode MS_EPI_POOL2) is built for this shape (16-byte aligned tensors assumed; bf16: 0 = fp32 entry point, 1 = `_bf16`, 2 = `_bf16m`)
extern "C" int ms_conv2d_pool2_ok(int N, int Cin, int H, int W, int Cout, int pro_mode, int bf16) {
  if (N < 1 || Cin < 1 || H < 2 || W < 4 || Cout < 1 || H % 2 != 0 || W % 4 != 0 || pro_mode < 0 || pro_mode > 2) return 0;
  ConvArgs a{};
  a.N = N; a.Cin = Cin; a.Hs = a.Hin = a.Hout = H; a.Ws = a.Win = a.Wout = W; a.Cout = a.cout_real = Cout;
  a.cin_pad = (Cin + 3) / 4 * 4; a.cout_pad = (Cout + 63) / 64 * 64;
  a.pro_mode = pro_mode; a.epi_mode = MS_EPI_POOL2; a.wino_ok = 1; a.act_bf16 = (bf16 < 0 || bf16 > 2) ? 2 : bf16;
  return (conv_wide_eligible(a, 3, 1, FETCH_NORMAL, true) && conv_wide_is_wino(a)) ? 1 : 0;
}
namespace ms { int conv_wino_blocks(const ConvArgs& a); bool conv_wino_blockform(const ConvArgs& a); bool conv_wino_flat(const ConvArgs& a); }      // ms_conv_inst_wino.hip / ms_conv_inst_winof.hip
namespace ms {
// (the arithmetic is wino_pack_one, ms_pack.h: shared with the batched appendix refresh ms_appendix_batch)
__global__ __launch_bounds__(256) void wino_pack_kernel(float* __restrict__ wp, int Cin, int Cout, int cin_pad, int cout_pad) {
  wino_pack_one((long)blockIdx.x * 256 + threadIdx.x, wp, Cin, Cout, cin_pad, cout_pad);
}
}  // namespace ms
extern "C" size_t ms_wino_pack_floats(int Cin, int Cout) {
  if (Cin < 8 || Cout < 1 || Cin % 8 != 0) return 0;
  return (size_t)((Cout + 15) / 16) * (size_t)(Cin / 8) * 2048;
}
extern "C" int ms_wino_pack(float* w_packed, int Cin, int Cout, void* stream) {
  if (w_packed == nullptr || ms_wino_pack_floats(Cin, Cout) == 0) { set_error("ms_wino_pack: packed 3x3 weights with Cin %% 8 == 0"); return MS_ERR_INVALID; }
  const int cin_pad = (Cin + 3) / 4 * 4, cout_pad = (Cout + 63) / 64 * 64;
  const long total = (long)((Cout + 15) / 16) * (Cin / 8) * 128;
  MS_LAUNCH(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_packed, Cin, Cout, cin_pad, cout_pad);
  return check_launch("wino_pack");
}
extern "C" int ms_conv2d_form(int N, int Cin, int H, int W, int Cout, int pro_mode, int bf16, int fetch) {
  if (N < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1 || pro_mode < 0 || pro_mode > 2 || fetch < 0 || (fetch & 0xFF) != 0) return 0;
  ConvArgs a{};
  a.N = N; a.Cin = Cin; a.Hs = a.Hin = a.Hout = H; a.Ws = a.Win = a.Wout = W; a.Cout = a.cout_real = Cout;
  a.cin_pad = (Cin + 3) / 4 * 4; a.cout_pad = (Cout + 63) / 64 * 64;
  a.pro_mode = pro_mode; a.wino_ok = (fetch & MS_FETCH_WINOGRAD) ? 1 : 0; a.wino_nt1 = (fetch & MS_FETCH_WINO_NT1) ? 1 : 0;
  a.wino_blocks = (fetch & MS_FETCH_WINO_BLOCKS) ? 1 : 0;
  a.act_bf16 = (bf16 < 0 || bf16 > 2) ? 2 : bf16;
  static const float appendix_marker = 0.f;             // (only compared with null by the dispatch rules)
  a.wu = ((fetch & MS_FETCH_WINO_U) && a.wino_ok && Cin % 8 == 0) ? &appendix_marker : nullptr;
  if (!conv_wide_eligible(a, 3, 1, FETCH_NORMAL, W % 4 == 0)) return conv_k3n_eligible(a, 3, 1, FETCH_NORMAL) ? 6 : 0;
  if (!conv_wide_is_wino(a)) return 1;
  if (conv_wino_flat(a)) return 7;                      // (the flat form needs the appendix and one of its epilogues: asked here with epi_mode 0)
  return 1 + conv_wino_blocks(a) + (conv_wino_blockform(a) ? 2 : 0);
}
// Channels a rider should have at most on a conv launch with this output shape so that every MFMA wave takes ONE channel: 4 x (an estimate of) the workgroup count.
// A speed hint for the caller's choice between a rider and a launch of its own - the kernel deals the channels round-robin over whatever grid it gets, so a launch
// whose grid turns out smaller (channel-block rounding, one workgroup per CU) still carries the whole job (this used to be a hard bound that ignored both).
extern "C" int ms_conv_ride_capacity(int N, int Hout, int Wout) {
  if (N < 1 || Hout < 1 || Wout < 1) return 0;
  const long tiles = (long)N * cdiv(Wout, tile_w(Wout)) * cdiv(Hout, tile_h(Wout));
  return (int)(4 * std::min<long>(tiles, (long)num_cus()));
}
extern "C" int ms_conv2d_ride(const float* in, const float* in2, float* out, const float* w_packed, const float* bias,
                              int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                              int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_nstride, int pro_cstride, float slope,
                              int epi_mode, float* stats, int ride_kind, const float* ride_tab, int ride_nparts, const float* ride_p0, const float* ride_p1, float ride_eps, double ride_count,
                              float* ride_out4, int ride_C,
                              void* stream) {
  const Ride rd{ride_kind, ride_tab, ride_nparts, ride_p0, ride_p1, ride_eps, ride_count, ride_out4, ride_C};
  return conv2d_impl(in, in2, out, w_packed, bias, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pro_mode, pro_a, pro_b, pro_c, pro_nstride, pro_cstride, slope,
                     epi_mode, stats, nullptr, stream, 0, nullptr, &rd);
}
extern "C" int ms_conv2d_ride_bf16(const uint16_t* in, const uint16_t* in2, uint16_t* out, const float* w_packed, const float* bias,
                                   int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                                   int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_nstride, int pro_cstride, float slope,
                                   int epi_mode, float* stats, int ride_kind, const float* ride_tab, int ride_nparts, const float* ride_p0, const float* ride_p1, float ride_eps, double ride_count,
                              float* ride_out4, int ride_C,
                                   void* stream) {
  const Ride rd{ride_kind, ride_tab, ride_nparts, ride_p0, ride_p1, ride_eps, ride_count, ride_out4, ride_C};
  return conv2d_impl(reinterpret_cast<const float*>(in), reinterpret_cast<const float*>(in2), reinterpret_cast<float*>(out), w_packed, bias, N, Cin, Hs, Ws, Cout, ks, stride, fetch,
                     pro_mode, pro_a, pro_b, pro_c, pro_nstride, pro_cstride, slope, epi_mode, stats, nullptr, stream, 1, nullptr, &rd);
}

extern "C" int ms_conv1x1_bnres(const float* in, float* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout,
                                const float* u, const float* coef4, float slope, int up2, void* stream) {
  MaskEpi mk{u, coef4, slope, nullptr};
  mk.mode = up2 ? 5 : 4;
  return conv2d_impl(in, nullptr, out, w_packed, bias, N, Cin, Hs, Ws, Cout, 1, 1, 0, 0, nullptr, nullptr, nullptr, 0, 1, 1.0f, 0, nullptr, &mk, stream);
}

extern "C" size_t ms_xfin_gran_bytes(int C) { return (size_t)C * kXfinNG * kXfinRep * sizeof(conv_u64_t); }

// ms_bn_finalize + ms_conv1x1_bnres in ONE launch: the BatchNorm coefficients of u's layer are derived inside this launch from the statistics table `stats`
// the conv that produced u wrote (one wave per channel, published to the waves that need them; see ConvArgs::xf_*), and stored to coef4 for later kernels.
extern "C" int ms_conv1x1_bnres_xfin(const float* in, float* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout,
                                     const float* u, const float* stats, const float* gamma, const float* beta, float eps, float* coef4, void* gran, int* err,
                                     float slope, int up2, void* stream) {
  MaskEpi mk{u, coef4, slope, nullptr};
  mk.mode = up2 ? 5 : 4;
  const XFin xf{stats, gamma, beta, eps, coef4, gran, err, Cout};
  return conv2d_impl(in, nullptr, out, w_packed, bias, N, Cin, Hs, Ws, Cout, 1, 1, 0, 0, nullptr, nullptr, nullptr, 0, 1, 1.0f, 0, nullptr, &mk, stream, 0, &xf);
}
extern "C" int ms_conv1x1_bnres_xfin_bf16(const uint16_t* in, uint16_t* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout,
                                          const uint16_t* u, const float* stats, const float* gamma, const float* beta, float eps, float* coef4, void* gran, int* err,
                                          float slope, int up2, void* stream) {
  MaskEpi mk{reinterpret_cast<const float*>(u), coef4, slope, nullptr};
  mk.mode = up2 ? 5 : 4;
  const XFin xf{stats, gamma, beta, eps, coef4, gran, err, Cout};
  return conv2d_impl(reinterpret_cast<const float*>(in), nullptr, reinterpret_cast<float*>(out), w_packed, bias, N, Cin, Hs, Ws, Cout, 1, 1, 0, 0, nullptr, nullptr, nullptr, 0, 1, 1.0f,
                     0, nullptr, &mk, stream, 1, &xf);
}

// ms_bn_finalize (kind 0, pro_mode 1) or ms_bn_bwd_coefs on a conv-epilogue table (kind 1, pro_mode 2) + the ms_conv2d that consumes the coefficients in its
// PROLOGUE, in one launch.  kind 0: tab = statistics table of the producing conv, p0 = gamma, p1 = beta, eps; kind 1: tab = the table of ms_conv2d_actbwd /
// ms_conv_subpix (float2 records with header), p0 = the forward records {sc, sh, mean, invstd} [Cin][4], count = N*H*W.  coef4 [Cin][4] receives the records.
static int conv2d_xfin_impl(const float* in, const float* in2, float* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride,
                            int fetch, int pro_mode, float slope, int epi_mode, float* stats, int kind, const float* tab, const float* p0, const float* p1, float eps, double count,
                            float* coef4, void* gran, int* err, void* stream, int act_bf16) {
  XFin xf{tab, p0, p1, eps, coef4, gran, err, Cin};
  xf.kind = kind; xf.count = count;
  return conv2d_impl(in, in2, out, w_packed, bias, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pro_mode, nullptr, nullptr, nullptr, 0, 4, slope, epi_mode, stats, nullptr, stream,
                     act_bf16, &xf);
}
extern "C" int ms_conv2d_xfin(const float* in, const float* in2, float* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride,
                              int fetch, int pro_mode, float slope, int epi_mode, float* stats, int kind, const float* tab, const float* p0, const float* p1, float eps, double count,
                              float* coef4, void* gran, int* err, void* stream) {
  return conv2d_xfin_impl(in, in2, out, w_packed, bias, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pro_mode, slope, epi_mode, stats, kind, tab, p0, p1, eps, count, coef4, gran, err, stream, 0);
}
extern "C" int ms_conv2d_xfin_bf16(const uint16_t* in, const uint16_t* in2, uint16_t* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride,
                                   int fetch, int pro_mode, float slope, int epi_mode, float* stats, int kind, const float* tab, const float* p0, const float* p1, float eps, double count,
                                   float* coef4, void* gran, int* err, void* stream) {
  return conv2d_xfin_impl(reinterpret_cast<const float*>(in), reinterpret_cast<const float*>(in2), reinterpret_cast<float*>(out), w_packed, bias, N, Cin, Hs, Ws, Cout, ks, stride, fetch,
                          pro_mode, slope, epi_mode, stats, kind, tab, p0, p1, eps, count, coef4, gran, err, stream, 1);
}

// ---- `_bf16` twins: the activation tensors (in, in2, out, u) hold bf16 bit patterns; weights, bias, coefficients, statistics and tables are fp32 as before.
// Built for the vector staging paths: rows of W % 4 == 0 elements (W % 2 for the fused up-sampling fetch), 16-byte aligned tensors.
static const float* as_f(const uint16_t* p) { return reinterpret_cast<const float*>(p); }
static float* as_f(uint16_t* p) { return reinterpret_cast<float*>(p); }
extern "C" int ms_conv2d_bf16(const uint16_t* in, const uint16_t* in2, uint16_t* out, const float* w_packed, const float* bias,
                              int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                              int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_nstride, int pro_cstride, float slope,
                              int epi_mode, float* stats, void* stream) {
  return conv2d_impl(as_f(in), as_f(in2), as_f(out), w_packed, bias, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pro_mode, pro_a, pro_b, pro_c, pro_nstride, pro_cstride, slope,
                     epi_mode, stats, nullptr, stream, 1);
}
extern "C" int ms_conv1x1_bnres_bf16(const uint16_t* in, uint16_t* out, const float* w_packed, const float* bias, int N, int Cin, int Hs, int Ws, int Cout,
                                     const uint16_t* u, const float* coef4, float slope, int up2, void* stream) {
  MaskEpi mk{as_f(u), coef4, slope, nullptr};
  mk.mode = up2 ? 5 : 4;
  return conv2d_impl(as_f(in), nullptr, as_f(out), w_packed, bias, N, Cin, Hs, Ws, Cout, 1, 1, 0, 0, nullptr, nullptr, nullptr, 0, 1, 1.0f, 0, nullptr, &mk, stream, 1);
}
extern "C" int ms_conv2d_actbwd_bf16(const uint16_t* in, const uint16_t* in2, uint16_t* out, const float* w_packed,
                                     int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                                     int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_nstride, int pro_cstride, float slope,
                                     const uint16_t* u, const float* coef4, float act_slope, float* tab, void* stream) {
  const MaskEpi mk{as_f(u), coef4, act_slope, tab};
  return conv2d_impl(as_f(in), as_f(in2), as_f(out), w_packed, nullptr, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pro_mode, pro_a, pro_b, pro_c, pro_nstride, pro_cstride, slope,
                     0, nullptr, &mk, stream, 1);
}

// ---- `_bf16m`: bf16 storage AND bf16 matrix arithmetic.  3x3 stride-1 convolutions with rows of >= 16 pixels run v_mfma_f32_16x16x16_bf16 (fp32 accumulation):
// their contraction operands - the prologue's output and the weights - are rounded to bf16 on the way into LDS; every other shape is ms_conv2d_bf16.
extern "C" int ms_conv2d_bf16m(const uint16_t* in, const uint16_t* in2, uint16_t* out, const float* w_packed, const float* bias,
                               int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                               int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_nstride, int pro_cstride, float slope,
                               int epi_mode, float* stats, void* stream) {
  return conv2d_impl(as_f(in), as_f(in2), as_f(out), w_packed, bias, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pro_mode, pro_a, pro_b, pro_c, pro_nstride, pro_cstride, slope,
                     epi_mode, stats, nullptr, stream, 2);
}
extern "C" int ms_conv2d_actbwd_bf16m(const uint16_t* in, const uint16_t* in2, uint16_t* out, const float* w_packed,
                                      int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                                      int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_nstride, int pro_cstride, float slope,
                                      const uint16_t* u, const float* coef4, float act_slope, float* tab, void* stream) {
  const MaskEpi mk{as_f(u), coef4, act_slope, tab};
  return conv2d_impl(as_f(in), as_f(in2), as_f(out), w_packed, nullptr, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pro_mode, pro_a, pro_b, pro_c, pro_nstride, pro_cstride, slope,
                     0, nullptr, &mk, stream, 2);
}

extern "C" size_t ms_conv_actbwd_tab_bytes(int Cout) { return ((size_t)Cout * kStatSlots + 1) * sizeof(float2); }

extern "C" int ms_conv2d_actbwd(const float* in, const float* in2, float* out, const float* w_packed,
                                int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                                int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_nstride, int pro_cstride, float slope,
                                const float* u, const float* coef4, float act_slope, float* tab, void* stream) {
  const MaskEpi mk{u, coef4, act_slope, tab};
  return conv2d_impl(in, in2, out, w_packed, nullptr, N, Cin, Hs, Ws, Cout, ks, stride, fetch, pro_mode, pro_a, pro_b, pro_c, pro_nstride, pro_cstride, slope,
                     0, nullptr, &mk, stream);
}

// ms_conv2d_actbwd whose two-tensor (BatchNorm-backward) PROLOGUE coefficients are derived inside the launch from the table of the activation-backward epilogue that
// produced `in` (`_xfin` kind 1: xf_tab = that table, xf_p0 = the forward records {sc, sh, mean, invstd} [Cin][4] of the layer being back-propagated, count = N*H*W):
// ms_bn_bwd_coefs + ms_conv2d_actbwd in one launch (round 5: the chains e.cd.da -> e.dz_i and e.d1.dx -> e.inc.da of the encoder's backward; the same bits in `out` and `tab`
// as the two launches, the records land in xf_coef4 for whoever reads them later).
static int conv2d_actbwd_xfin_impl(const float* in, const float* in2, float* out, const float* w_packed, int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                                   const float* u, const float* coef4, float act_slope, float* tab, const float* xf_tab, const float* xf_p0, double count, float* xf_coef4,
                                   void* gran, int* err, void* stream, int act_bf16) {
  const MaskEpi mk{u, coef4, act_slope, tab};
  XFin xf{xf_tab, xf_p0, nullptr, 0.f, xf_coef4, gran, err, Cin};
  xf.kind = 1; xf.count = count;
  return conv2d_impl(in, in2, out, w_packed, nullptr, N, Cin, Hs, Ws, Cout, ks, stride, fetch, 2, nullptr, nullptr, nullptr, 0, 4, 1.0f, 0, nullptr, &mk, stream, act_bf16, &xf);
}
extern "C" int ms_conv2d_actbwd_xfin(const float* in, const float* in2, float* out, const float* w_packed, int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                                     const float* u, const float* coef4, float act_slope, float* tab, const float* xf_tab, const float* xf_p0, double count, float* xf_coef4,
                                     void* gran, int* err, void* stream) {
  return conv2d_actbwd_xfin_impl(in, in2, out, w_packed, N, Cin, Hs, Ws, Cout, ks, stride, fetch, u, coef4, act_slope, tab, xf_tab, xf_p0, count, xf_coef4, gran, err, stream, 0);
}
extern "C" int ms_conv2d_actbwd_xfin_bf16(const uint16_t* in, const uint16_t* in2, uint16_t* out, const float* w_packed, int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride,
                                          int fetch, const uint16_t* u, const float* coef4, float act_slope, float* tab, const float* xf_tab, const float* xf_p0, double count,
                                          float* xf_coef4, void* gran, int* err, void* stream) {
  return conv2d_actbwd_xfin_impl(as_f(in), as_f(in2), as_f(out), w_packed, N, Cin, Hs, Ws, Cout, ks, stride, fetch, as_f(u), coef4, act_slope, tab, xf_tab, xf_p0, count, xf_coef4, gran,
                                 err, stream, 1);
}

extern "C" int ms_bn_finalize(const float* stats, int nparts, const float* gamma, const float* beta, float eps, float* coef4, int C, void* stream) {
  if (C < 1 || nparts != kStatSlots) { set_error("ms_bn_finalize: invalid shape (nparts must be ms_conv_stats_parts())"); return MS_ERR_INVALID; }
  MS_LAUNCH(bn_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, (const float4*)stats, gamma, beta, eps, (float4*)coef4);
  return check_launch("bn_finalize");
}
