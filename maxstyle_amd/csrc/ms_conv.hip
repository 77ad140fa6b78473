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
#include "ms_common.h"
#include "maxstyle_hip.h"

namespace ms {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { FETCH_NORMAL = 0, FETCH_UPS2 = 1, FETCH_ZINS2 = 2 };

struct ConvArgs {
  const float* in; const float* in2; float* out; const float* w; const float* bias;
  const float* pro_a; const float* pro_b; const float* pro_c;
  float* stats;                 // PlanePartial-like float4 [cout][nparts] or null
  int N, Cin, Hs, Ws, Hin, Win, Cout, Hout, Wout, cin_pad, cout_pad;
  int pro_mode, pro_nstride, pro_cstride; float slope;
  int epi_mode, tiles_x, tiles_y, cout_real;   // cout_real: ConvTranspose real channel count (epi shuffle)
};

constexpr int TH = 8, TW = 32;

template <int KS, int STRIDE>
struct Geo {
  static constexpr int PAD = (KS == 3) ? 1 : 0;
  static constexpr int IH = (TH - 1) * STRIDE + KS;
  static constexpr int IW = (TW - 1) * STRIDE + KS;
  static constexpr int HALFW = (IW + 1) / 2;
  static constexpr int RS = (STRIDE == 1) ? ((IW + 3) / 4 * 4) : 2 * HALFW;
  static constexpr int BASE = IH * RS;
  static constexpr int PS = BASE + ((16 - BASE % 32 + 32) % 32);   // plane stride == 16 (mod 32 banks)
  static constexpr int CK = (STRIDE == 1) ? 16 : 8;
};

template <int NT> struct WGeo { static constexpr int WS = (NT == 1) ? 16 : NT * 16 + 16; };

template <int KS, int STRIDE, int FETCH, int NT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
  using G = Geo<KS, STRIDE>;
  constexpr int CK = G::CK, PS = G::PS, RS = G::RS, IH = G::IH, IW = G::IW, PAD = G::PAD, HALFW = G::HALFW;
  constexpr int WS = WGeo<NT>::WS;
  constexpr int TAPS = KS * KS;
  constexpr int COUT_TILE = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* in_lds = smem;                 // [CK][PS]
  float* w_lds = smem + CK * PS;        // [TAPS][CK][WS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, k = lane >> 4;
  const int tile = blockIdx.x;
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int co0 = blockIdx.y * COUT_TILE;
  const int n = blockIdx.z;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const size_t in_plane = (size_t)a.Hs * a.Ws;
  const float* in_n = a.in + (size_t)n * a.Cin * in_plane;
  const float* in2_n = a.in2 ? a.in2 + (size_t)n * a.Cin * in_plane : nullptr;

  const int a_lane = k * PS + (wave * 2 * STRIDE) * RS + m;
  const int b_lane = k * WS + m;

  for (int c0 = 0; c0 < a.cin_pad; c0 += CK) {
    __syncthreads();   // previous chunk's readers are done
    // ---- stage the input halo tile (prologue applied on the fly) ----
    for (int idx = tid; idx < CK * IH * IW; idx += 256) {
      const int c = idx / (IH * IW);
      const int rem = idx - c * (IH * IW);
      const int r = rem / IW;
      const int xx = rem - r * IW;
      const int Y = iy0 + r, X = ix0 + xx;
      const int ci = c0 + c;
      bool ok = (ci < a.Cin) && (Y >= 0) && (Y < a.Hin) && (X >= 0) && (X < a.Win);
      int ys = Y, xs = X;
      if (FETCH == FETCH_UPS2) { ys = Y >> 1; xs = X >> 1; }
      if (FETCH == FETCH_ZINS2) { ok = ok && !((Y | X) & 1); ys = Y >> 1; xs = X >> 1; }
      float v = 0.f;
      if (ok) {
        const size_t off = (size_t)ci * in_plane + (size_t)ys * a.Ws + xs;
        v = in_n[off];
        if (a.pro_mode == 1) {
          const int pi = (n * a.pro_nstride + ci) * a.pro_cstride;
          v = leaky(a.pro_a[pi] * v + a.pro_b[pi], a.slope);
        } else if (a.pro_mode == 2) {
          const int pi = ci * a.pro_cstride;
          v = a.pro_a[pi] * v + a.pro_b[pi] * in2_n[off] + a.pro_c[pi];
        }
      }
      const int q = (STRIDE == 1) ? xx : ((xx & 1) * HALFW + (xx >> 1));
      in_lds[c * PS + r * RS + q] = v;
    }
    // ---- stage the weight slice [TAPS][CK][COUT_TILE] ----
    for (int idx = tid; idx < TAPS * CK * (COUT_TILE / 4); idx += 256) {
      const int j4 = idx % (COUT_TILE / 4);
      const int row = idx / (COUT_TILE / 4);      // tap*CK + c
      const int c = row % CK, tap = row / CK;
      float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 + c < a.cin_pad)
        wv = *reinterpret_cast<const float4*>(a.w + ((size_t)tap * a.cin_pad + c0 + c) * a.cout_pad + co0 + j4 * 4);
      *reinterpret_cast<float4*>(w_lds + row * WS + j4 * 4) = wv;
    }
    __syncthreads();
    const int ncg = min(CK / 4, (a.cin_pad - c0) / 4);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int ky = tap / KS, kx = tap % KS;
      const int tap_off = ky * RS + ((STRIDE == 1) ? kx : ((kx & 1) * HALFW + (kx >> 1)));
      for (int cg = 0; cg < ncg; ++cg) {
        float bf[NT], af[4];
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = w_lds[b_lane + (tap * CK + cg * 4) * WS + j * 16];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = in_lds[a_lane + cg * 4 * PS + tap_off + (i >> 1) * STRIDE * RS + (i & 1) * 16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue ----
  // D layout (16x16): column (output channel) = lane&15, rows (pixels) = 4*(lane>>4) + reg
  const int xq = 4 * k;     // first of this lane's 4 consecutive pixels inside a 16-pixel M-tile
  float bias_v[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = co0 + j * 16 + m;
    int bidx = co;
    if (a.epi_mode == 2) bidx = co % a.cout_real;
    bias_v[j] = (a.bias != nullptr && co < ((a.epi_mode == 2) ? 4 * a.cout_real : a.Cout)) ? a.bias[bidx] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] += bias_v[j];

  if (a.stats != nullptr) {
    // per-channel block statistics of the outputs: (count, mean, M2), two-pass from registers
    __syncthreads();
    float* red = smem;                       // [4 waves][COUT_TILE]
    float s[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) s[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = oy0 + wave * 2 + (i >> 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = ox0 + (i & 1) * 16 + xq + r;
        const bool ok = (y < a.Hout) && (x < a.Wout);
#pragma unroll
        for (int j = 0; j < NT; ++j) s[j] += ok ? acc[i][j][r] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      s[j] += __shfl_xor(s[j], 16, 64);
      s[j] += __shfl_xor(s[j], 32, 64);
      if (k == 0) red[wave * COUT_TILE + j * 16 + m] = s[j];
    }
    __syncthreads();
    const float cnt = (float)(min(TH, a.Hout - oy0) * min(TW, a.Wout - ox0));
    float mean[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int cc = j * 16 + m;
      mean[j] = (red[cc] + red[COUT_TILE + cc] + red[2 * COUT_TILE + cc] + red[3 * COUT_TILE + cc]) / cnt;
      s[j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = oy0 + wave * 2 + (i >> 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = ox0 + (i & 1) * 16 + xq + r;
        const bool ok = (y < a.Hout) && (x < a.Wout);
#pragma unroll
        for (int j = 0; j < NT; ++j) { const float d = acc[i][j][r] - mean[j]; s[j] += ok ? d * d : 0.f; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      s[j] += __shfl_xor(s[j], 16, 64);
      s[j] += __shfl_xor(s[j], 32, 64);
      if (k == 0) red[wave * COUT_TILE + j * 16 + m] = s[j];
    }
    __syncthreads();
    if (wave == 0 && k == 0) {
      const int nparts = a.N * a.tiles_x * a.tiles_y;
      const int pidx = n * (a.tiles_x * a.tiles_y) + tile;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int cc = j * 16 + m;
        const int co = co0 + cc;
        if (co < a.Cout) {
          const float m2 = red[cc] + red[COUT_TILE + cc] + red[2 * COUT_TILE + cc] + red[3 * COUT_TILE + cc];
          reinterpret_cast<float4*>(a.stats)[(size_t)co * nparts + pidx] = make_float4(cnt, mean[j], m2, 0.f);
        }
      }
    }
  }

  if (a.epi_mode == 2) {
    // ConvTranspose2d k=2 s=2: GEMM column j = (dy*2+dx)*cout_real + co -> out[n,co,2y+dy,2x+dx]
    const int Ho = 2 * a.Hout, Wo = 2 * a.Wout;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = co0 + j * 16 + m;
      if (col >= 4 * a.cout_real) continue;
      const int q = col / a.cout_real, co = col - q * a.cout_real;
      const int dy = q >> 1, dx = q & 1;
      float* op = a.out + ((size_t)n * a.cout_real + co) * Ho * Wo;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int y = oy0 + wave * 2 + (i >> 1);
        if (y >= a.Hout) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int x = ox0 + (i & 1) * 16 + xq + r;
          if (x < a.Wout) op[(size_t)(2 * y + dy) * Wo + 2 * x + dx] = acc[i][j][r];
        }
      }
    }
    return;
  }

  const bool vec_ok = (a.Wout % 4 == 0);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = co0 + j * 16 + m;
    if (co >= a.Cout) continue;
    float* op = a.out + ((size_t)n * a.Cout + co) * a.Hout * a.Wout;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = oy0 + wave * 2 + (i >> 1);
      if (y >= a.Hout) continue;
      const int x = ox0 + (i & 1) * 16 + xq;
      float* o = op + (size_t)y * a.Wout + x;
      if (vec_ok && x + 3 < a.Wout) {
        float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (a.epi_mode == 1) { const float4 p = *reinterpret_cast<const float4*>(o); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
        *reinterpret_cast<float4*>(o) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (x + r < a.Wout) o[r] = (a.epi_mode == 1) ? (o[r] + acc[i][j][r]) : acc[i][j][r];
      }
    }
  }
}

template <int KS, int STRIDE, int FETCH, int NT>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
  using G = Geo<KS, STRIDE>;
  constexpr int lds_floats = G::CK * G::PS + KS * KS * G::CK * WGeo<NT>::WS;
  constexpr int red_floats = 4 * 16 * NT;
  constexpr size_t lds_bytes = sizeof(float) * (lds_floats > red_floats ? lds_floats : red_floats);
  static bool attr_set = false;
  if (!attr_set && lds_bytes > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_mfma_kernel<KS, STRIDE, FETCH, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_set = true;
  }
  const int gemm_cols = (a.epi_mode == 2) ? 4 * a.cout_real : a.Cout;
  dim3 grid(a.tiles_x * a.tiles_y, cdiv(gemm_cols, 16 * NT), a.N), block(256);
  hipLaunchKernelGGL((conv_mfma_kernel<KS, STRIDE, FETCH, NT>), grid, block, lds_bytes, st, a);
  return check_launch("conv_mfma");
}

template <int KS, int STRIDE, int FETCH>
static int launch_conv_nt(const ConvArgs& a, int nt, hipStream_t st) {
  switch (nt) {
    case 1: return launch_conv<KS, STRIDE, FETCH, 1>(a, st);
    case 2: return launch_conv<KS, STRIDE, FETCH, 2>(a, st);
    default: return launch_conv<KS, STRIDE, FETCH, 4>(a, st);
  }
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm statistics finalize: Chan-merge the per-block (count, mean, M2) in fp64 -> per channel
//   out[c] = {scale = gamma*invstd, shift = beta - mean*scale, mean, invstd}      (biased variance, eps)
// model_util.py:468-510 mode: batch statistics, running buffers untouched.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float4* __restrict__ part, int nparts, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float4* __restrict__ out) {
  __shared__ double sn[256], sm[256], sq[256];
  const int c = blockIdx.x;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
    const float4 q = part[(size_t)c * nparts + i];
    chan_merge(n, mean, m2, (double)q.x, (double)q.y, (double)q.z);
  }
  sn[threadIdx.x] = n; sm[threadIdx.x] = mean; sq[threadIdx.x] = m2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      double n1 = sn[threadIdx.x], m1 = sm[threadIdx.x], q1 = sq[threadIdx.x];
      chan_merge(n1, m1, q1, sn[threadIdx.x + off], sm[threadIdx.x + off], sq[threadIdx.x + off]);
      sn[threadIdx.x] = n1; sm[threadIdx.x] = m1; sq[threadIdx.x] = q1;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double var = sq[0] / sn[0];
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    out[c] = make_float4(sc, beta[c] - (float)sm[0] * sc, (float)sm[0], invstd);
  }
}

}  // namespace ms

using namespace ms;

extern "C" size_t ms_conv_stats_bytes(int N, int Cout, int Hout, int Wout) {
  return (size_t)Cout * N * cdiv(Hout, TH) * cdiv(Wout, TW) * sizeof(float4);
}

extern "C" int ms_conv_stats_parts(int N, int Hout, int Wout) { return N * cdiv(Hout, TH) * cdiv(Wout, TW); }

extern "C" int ms_conv2d(const float* in, const float* in2, float* out, const float* w_packed, const float* bias,
                         int N, int Cin, int Hs, int Ws, int Cout, int ks, int stride, int fetch,
                         int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_nstride, int pro_cstride, float slope,
                         int epi_mode, float* stats, void* stream) {
  if (N < 1 || Cin < 1 || Cout < 1 || Hs < 1 || Ws < 1) { set_error("ms_conv2d: invalid shape"); return MS_ERR_INVALID; }
  if (pro_mode < 0 || pro_mode > 2 || epi_mode < 0 || epi_mode > 2 || fetch < 0 || fetch > 2) { set_error("ms_conv2d: invalid mode"); return MS_ERR_INVALID; }
  if (pro_mode == 2 && in2 == nullptr) { set_error("ms_conv2d: pro_mode 2 needs in2"); return MS_ERR_INVALID; }
  if (pro_mode != 0 && (pro_a == nullptr || pro_b == nullptr)) { set_error("ms_conv2d: prologue coefficients missing"); return MS_ERR_INVALID; }
  if (epi_mode == 2 && (ks != 1 || stride != 1 || fetch != 0 || stats != nullptr)) { set_error("ms_conv2d: pixel-shuffle epilogue is for the k=1 GEMM form of ConvTranspose2d(k=2,s=2)"); return MS_ERR_INVALID; }
  if (!aligned16(w_packed)) { set_error("ms_conv2d: packed weights must be 16-byte aligned"); return MS_ERR_ALIGN; }
  ConvArgs a{};
  a.in = in; a.in2 = in2; a.out = out; a.w = w_packed; a.bias = bias;
  a.pro_a = pro_a; a.pro_b = pro_b; a.pro_c = pro_c; a.stats = stats;
  a.N = N; a.Cin = Cin; a.Hs = Hs; a.Ws = Ws;
  a.Hin = (fetch == FETCH_NORMAL) ? Hs : 2 * Hs; a.Win = (fetch == FETCH_NORMAL) ? Ws : 2 * Ws;
  const int pad = (ks == 3) ? 1 : 0;
  a.Hout = (a.Hin + 2 * pad - ks) / stride + 1; a.Wout = (a.Win + 2 * pad - ks) / stride + 1;
  a.cout_real = Cout;
  const int gemm_cols = (epi_mode == 2) ? 4 * Cout : Cout;
  a.Cout = (epi_mode == 2) ? gemm_cols : Cout;
  a.cin_pad = (Cin + 3) / 4 * 4; a.cout_pad = (gemm_cols + 63) / 64 * 64;
  a.pro_mode = pro_mode; a.pro_nstride = pro_nstride; a.pro_cstride = pro_cstride < 1 ? 1 : pro_cstride; a.slope = slope; a.epi_mode = epi_mode;
  a.tiles_x = cdiv(a.Wout, TW); a.tiles_y = cdiv(a.Hout, TH);
  if (a.Hout < 1 || a.Wout < 1) { set_error("ms_conv2d: empty output"); return MS_ERR_INVALID; }
  if ((long)N > 65535 ) { set_error("ms_conv2d: batch too large for gridDim.z"); return MS_ERR_INVALID; }
  // output-channel tile: widest that still gives >= ~512 workgroups (256 CUs x 2)
  const long tiles = (long)a.tiles_x * a.tiles_y * N;
  int nt = 4;
  while (nt > 1 && (gemm_cols <= 16 * (nt / 2) || tiles * cdiv(gemm_cols, 16 * nt) < 512)) nt >>= 1;
  hipStream_t st = (hipStream_t)stream;
#define MS_CONV_CASE(K, S, F) if (ks == K && stride == S && fetch == F) return launch_conv_nt<K, S, F>(a, nt, st)
  MS_CONV_CASE(3, 1, FETCH_NORMAL);
  MS_CONV_CASE(3, 1, FETCH_UPS2);
  MS_CONV_CASE(3, 1, FETCH_ZINS2);
  MS_CONV_CASE(3, 2, FETCH_NORMAL);
  MS_CONV_CASE(1, 1, FETCH_NORMAL);
  MS_CONV_CASE(2, 2, FETCH_NORMAL);
#undef MS_CONV_CASE
  set_error("ms_conv2d: unsupported (ks=%d, stride=%d, fetch=%d)", ks, stride, fetch);
  return MS_ERR_INVALID;
}

extern "C" int ms_bn_finalize(const float* stats, int nparts, const float* gamma, const float* beta, float eps, float* coef4, int C, void* stream) {
  if (C < 1 || nparts < 1) { set_error("ms_bn_finalize: invalid shape"); return MS_ERR_INVALID; }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, (const float4*)stats, nparts, gamma, beta, eps, (float4*)coef4);
  return check_launch("bn_finalize");
}
