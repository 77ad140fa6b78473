// MaxStyle layer kernels for gfx950 (K1 forward, K2 backward, Adam on the style parameters).
//
// Reference semantics: /root/reference/src/advanced/maxstyle.py:140-189 (forward), autograd of it
// (SURVEY.md A.2), torch.optim.Adam defaults.
//
// Data layout: x is NCHW fp32 contiguous; a "plane" is one (b,c) image of HW floats.  All kernels are
// HBM-streaming: 16 B/lane coalesced loads, per-wave __shfl reductions, LDS only for the cross-wave
// combine, fixed-order (deterministic) two-stage reductions, no float atomics.
//
//  forward  = moments_partial (1 read of x; register-resident two-pass variance per chunk)
//           -> style_finalize (per channel: Chan-merge partials in fp64, batch std, mixing -> per-plane A,S)
//           -> restyle        (read x, write y = A*(x-mu)/sig + S)
//    12 B/element of traffic for 8 algorithmic (the second read of x is served from L2/Infinity Cache when the tensor fits);
//    ms_style_fwd dispatches to the single-read kernel of ms_style_fused.hip whenever the shape is eligible
//  backward = restyle_bwd (read dy,x; write dx=dy*A/sig; partial S1=sum dy, S2=sum dy*xhat)
//           -> style_bwd_finalize (d gamma, d beta, d lmda)
#include <algorithm>
#include <cstdlib>
#include "ms_common.h"
#include "maxstyle_hip.h"

namespace ms {

constexpr int kStyleThreads = 256;

struct PlanePartial { float n, mean, m2, pad; };

// ---------------------------------------------------------------------------------------------
// moments: grid (S, P). Block s of plane p owns elements [s*chunk, min(HW,(s+1)*chunk)).
// VEC=4: float4 loads (needs HW%4==0 and 16-B aligned base); VEC=1: scalar loads (any shape).
// NV = values of width VEC held per thread: chunk <= 256*VEC*NV.
// ---------------------------------------------------------------------------------------------
template <int VEC, int NV>
__global__ __launch_bounds__(kStyleThreads) void moments_partial_kernel(const float* __restrict__ x, PlanePartial* __restrict__ part,
                                                                       int HW, int chunk, int S) {
  __shared__ float red[16];
  const int p = blockIdx.y, s = blockIdx.x;
  const int beg = s * chunk;
  const int end = min(HW, beg + chunk);
  const float* xp = x + (size_t)p * HW;
  float v[NV][VEC];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = beg + (j * kStyleThreads + threadIdx.x) * VEC;
    if (VEC == 4) {
      if (i < end) {
        const float4 t = *reinterpret_cast<const float4*>(xp + i);
        v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
      } else {
        v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
      }
    } else {
      v[j][0] = (i < end) ? xp[i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) sum += v[j][k];
  }
  const float n = (float)(end - beg);
  const float mean = block_sum(sum, red) / n;
  float m2 = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = beg + (j * kStyleThreads + threadIdx.x) * VEC;
    if (i < end) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) { const float d = v[j][k] - mean; m2 += d * d; }
    }
  }
  m2 = block_sum(m2, red);
  if (threadIdx.x == 0) part[(size_t)p * S + s] = PlanePartial{n, mean, m2, 0.f};
}

__device__ __forceinline__ void merge_plane(const PlanePartial* part, int p, int S, int HW, float eps, float& mu, float& sig) {
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int s = 0; s < S; ++s) {
    const PlanePartial q = part[(size_t)p * S + s];
    chan_merge(n, mean, m2, (double)q.n, (double)q.mean, (double)q.m2);
  }
  mu = (float)mean;
  const float var = (float)(m2 / (double)(HW - 1));   // unbiased, as torch.var default
  sig = sqrtf(var + eps);
}

__global__ void moments_merge_kernel(const PlanePartial* __restrict__ part, float* __restrict__ mu, float* __restrict__ sig,
                                     int P, int S, int HW, float eps) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float m, sg;
  merge_plane(part, p, S, HW, eps, m, sg);
  mu[p] = m; sig[p] = sg;
}

// ---------------------------------------------------------------------------------------------
// finalize: one block per channel c, thread b handles plane (b,c).
//   merge partials -> mu,sig ; (first call) gamma_std[c]=std_b(sig), beta_std[c]=std_b(mu) (unbiased, fp64)
//   lam=clamp(lmda[b],0,1); A = sig(1-lam)+sig[perm b]lam + gamma_noise*gamma_std ; S likewise with mu/beta.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void style_finalize_kernel(const PlanePartial* __restrict__ part, float* __restrict__ mu, float* __restrict__ sig,
                                                              float* __restrict__ gamma_std, float* __restrict__ beta_std, int compute_std,
                                                              const float* __restrict__ lmda, const float* __restrict__ gamma_noise,
                                                              const float* __restrict__ beta_noise, const int64_t* __restrict__ perm,
                                                              float* __restrict__ coefA, float* __restrict__ coefS,
                                                              int B, int C, int S, int HW, float eps) {
  __shared__ double redd[16];
  extern __shared__ float sm[];     // [2*B]: mu_b, sig_b of this channel
  float* smu = sm; float* ssig = sm + B;
  const int c = blockIdx.x;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float m, sg;
    if (part != nullptr) {
      merge_plane(part, b * C + c, S, HW, eps, m, sg);
      mu[b * C + c] = m; sig[b * C + c] = sg;
    } else {
      m = mu[b * C + c]; sg = sig[b * C + c];
    }
    smu[b] = m; ssig[b] = sg;
  }
  __syncthreads();
  float gs, bs;
  if (compute_std & 1) {
    double am = 0.0, as = 0.0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) { am += (double)smu[b]; as += (double)ssig[b]; }
    const double mean_mu = block_sum_d(am, redd) / B;
    const double mean_sg = block_sum_d(as, redd) / B;
    double qm = 0.0, qs = 0.0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
      const double d1 = (double)smu[b] - mean_mu, d2 = (double)ssig[b] - mean_sg;
      qm += d1 * d1; qs += d2 * d2;
    }
    qm = block_sum_d(qm, redd); qs = block_sum_d(qs, redd);
    bs = (float)sqrt(qm / (double)(B - 1));
    gs = (float)sqrt(qs / (double)(B - 1));
    if (threadIdx.x == 0) { gamma_std[c] = gs; beta_std[c] = bs; }
  } else {
    gs = gamma_std[c]; bs = beta_std[c];
  }
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float m = smu[b], sg = ssig[b];
    float A = sg, Sh = m;
    if (lmda != nullptr) {
      const float lam = (compute_std & 2) ? lmda[b] : fminf(fmaxf(lmda[b], 0.f), 1.f);     // bit 1: MixStyle (no clamp)
      const int pb = (int)perm[b];
      A = sg * (1.f - lam) + ssig[pb] * lam;
      Sh = m * (1.f - lam) + smu[pb] * lam;
    }
    if (gamma_noise != nullptr) {
      A += gamma_noise[b * C + c] * gs;
      Sh += beta_noise[b * C + c] * bs;
    }
    coefA[b * C + c] = A;
    coefS[b * C + c] = Sh;
  }
}

// y = A * ((x - mu) / sig) + S, grid (S, P) like the moments kernel.
template <int VEC>
__global__ __launch_bounds__(kStyleThreads) void restyle_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               const float* __restrict__ mu, const float* __restrict__ sig,
                                                               const float* __restrict__ coefA, const float* __restrict__ coefS,
                                                               int HW, int chunk) {
  const int p = blockIdx.y;
  const float m = mu[p], a = coefA[p] / sig[p], sh = coefS[p];
  const int beg = blockIdx.x * chunk, end = min(HW, beg + chunk);
  const float* xp = x + (size_t)p * HW;
  float* yp = y + (size_t)p * HW;
  for (int i = beg + threadIdx.x * VEC; i < end; i += kStyleThreads * VEC) {
    if (VEC == 4) {
      float4 t = *reinterpret_cast<const float4*>(xp + i);
      t.x = a * (t.x - m) + sh; t.y = a * (t.y - m) + sh; t.z = a * (t.z - m) + sh; t.w = a * (t.w - m) + sh;
      *reinterpret_cast<float4*>(yp + i) = t;
    } else {
      yp[i] = a * (xp[i] - m) + sh;
    }
  }
}

// backward: partial S1 = sum dy, S2 = sum dy*xhat; dx = dy * A/sig (optional)
// bn_u != nullptr (ms_style_bwd_actbwd): x is the OUTPUT of a residual block (the layer sits right behind it), so the gradient this kernel hands back goes
// straight into that block's output-activation backward: dx is additionally multiplied by lrelu'(x) (sign(x) == sign(pre-activation)) and the two sums the
// BatchNorm backward of the block's last BatchNorm needs (sum g', sum g'*(u - mean_c), u = that BatchNorm's raw input) are written per block - what
// ms_act_bwd_reduce would do in its own 4-pass launch over dx.  The style gradients use the UNMASKED dy, as before.
// the head in front of a MaxStyle layer's backward (ms_style_bwd_head): g [N,K,HW] gradient w.r.t. the head's output, out = its sigmoid output (NULL: no sigmoid),
// w [K][C] the 1x1 head weights
struct HeadSrc { const float* g; const float* out; const float* w; int K; };
template <int VEC>
__global__ __launch_bounds__(kStyleThreads) void restyle_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                                                   const float* __restrict__ mu, const float* __restrict__ sig,
                                                                   const float* __restrict__ coefA, float2* __restrict__ part,
                                                                   int HW, int chunk, int S,
                                                                   const float* __restrict__ bn_u, const float4* __restrict__ bn_coef, float2* __restrict__ bn_part,
                                                                   int C, float slope, const HeadSrc hd) {
  __shared__ float red[16];
  const int p = blockIdx.y;
  const float m = mu[p], inv = 1.f / sig[p], a = coefA[p] * inv;
  const int beg = blockIdx.x * chunk, end = min(HW, beg + chunk);
  const float* xp = x + (size_t)p * HW;
  const float* gp = hd.g ? nullptr : dy + (size_t)p * HW;
  // hd.g != NULL: dy is not read - it is the gradient that ms_head_bwd would have written for this plane, formed on the fly from the K planes of the head's
  // output gradient: dy[c] = sum_k w[k][c] * (g_k * out_k * (1 - out_k)), the same operations in the same order (bit-identical to the stored tensor)
  const int hn = p / C, hc = p % C;
  float hw[4] = {0.f, 0.f, 0.f, 0.f};
  if (hd.g) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < hd.K) hw[k] = hd.w[k * C + hc];
  }
  auto head_dy4 = [&](int i) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < hd.K) {
      const size_t off = ((size_t)hn * hd.K + k) * HW + i;
      const float4 g = *reinterpret_cast<const float4*>(hd.g + off);
      float4 d = g;
      if (hd.out) { const float4 o = *reinterpret_cast<const float4*>(hd.out + off); d = make_float4(g.x * (o.x * (1.f - o.x)), g.y * (o.y * (1.f - o.y)), g.z * (o.z * (1.f - o.z)), g.w * (o.w * (1.f - o.w))); }
      acc.x += hw[k] * d.x; acc.y += hw[k] * d.y; acc.z += hw[k] * d.z; acc.w += hw[k] * d.w;
    }
    return acc;
  };
  float* dxp = dx ? dx + (size_t)p * HW : nullptr;
  const float* up = bn_u ? bn_u + (size_t)p * HW : nullptr;
  const float bmean = bn_u ? bn_coef[p % C].z : 0.f;
  float s1 = 0.f, s2 = 0.f, b1 = 0.f, b2 = 0.f;
  for (int i = beg + threadIdx.x * VEC; i < end; i += kStyleThreads * VEC) {
    if (VEC == 4) {
      const float4 g = hd.g ? head_dy4(i) : *reinterpret_cast<const float4*>(gp + i);
      const float4 t = *reinterpret_cast<const float4*>(xp + i);
      s1 += (g.x + g.y) + (g.z + g.w);
      s2 += g.x * ((t.x - m) * inv) + g.y * ((t.y - m) * inv) + g.z * ((t.z - m) * inv) + g.w * ((t.w - m) * inv);
      float4 o = make_float4(g.x * a, g.y * a, g.z * a, g.w * a);
      if (up) {
        const float4 uu = *reinterpret_cast<const float4*>(up + i);
        o.x *= (t.x > 0.f) ? 1.f : slope; o.y *= (t.y > 0.f) ? 1.f : slope; o.z *= (t.z > 0.f) ? 1.f : slope; o.w *= (t.w > 0.f) ? 1.f : slope;
        b1 += (o.x + o.y) + (o.z + o.w);
        b2 += (o.x * (uu.x - bmean) + o.y * (uu.y - bmean)) + (o.z * (uu.z - bmean) + o.w * (uu.w - bmean));
      }
      if (dxp) *reinterpret_cast<float4*>(dxp + i) = o;
    } else {
      const float g = gp[i];
      s1 += g; s2 += g * ((xp[i] - m) * inv);
      float o = g * a;
      if (up) { o *= (xp[i] > 0.f) ? 1.f : slope; b1 += o; b2 += o * (up[i] - bmean); }
      if (dxp) dxp[i] = o;
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) part[(size_t)p * S + blockIdx.x] = make_float2(s1, s2);
  if (up) {
    b1 = block_sum(b1, red);
    b2 = block_sum(b2, red);
    const int c = p % C, n = p / C, N = (int)gridDim.y / C;
    if (threadIdx.x == 0) bn_part[(size_t)c * (N * S) + n * S + blockIdx.x] = make_float2(b1, b2);      // the layout of ms_act_bwd_reduce's partials
  }
}

// one block per batch sample b; threads over channels.
__global__ __launch_bounds__(256) void style_bwd_finalize_kernel(const float2* __restrict__ part, const float* __restrict__ mu, const float* __restrict__ sig,
                                                                  const float* __restrict__ gamma_std, const float* __restrict__ beta_std,
                                                                  const float* __restrict__ lmda, const int64_t* __restrict__ perm,
                                                                  float* __restrict__ d_gamma, float* __restrict__ d_beta, float* __restrict__ d_lmda,
                                                                  int B, int C, int S) {
  __shared__ double redd[16];
  const int b = blockIdx.x;
  const int pb = perm ? (int)perm[b] : b;
  double acc = 0.0;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int p = b * C + c;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < S; ++s) { const float2 q = part[(size_t)p * S + s]; s1 += (double)q.x; s2 += (double)q.y; }
    if (d_gamma) d_gamma[p] = (float)((double)gamma_std[c] * s2);
    if (d_beta) d_beta[p] = (float)((double)beta_std[c] * s1);
    const int q = pb * C + c;
    acc += ((double)sig[q] - (double)sig[p]) * s2 + ((double)mu[q] - (double)mu[p]) * s1;
  }
  acc = block_sum_d(acc, redd);
  if (threadIdx.x == 0 && d_lmda) {
    const float l = lmda[b];
    d_lmda[b] = (l >= 0.f && l <= 1.f) ? (float)acc : 0.f;   // clamp(): zero gradient outside [0,1]
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 activation storage (`*_bf16` entry points): x / y / dy / dx are bf16 (16-byte accesses = 8 values), every statistic, coefficient and
// partial sum stays fp32 / fp64 and the finalize kernels above are shared.  Vector path only: H*W % 8 == 0, 16-byte aligned tensors.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bf16x8_load(const uint16_t* p, float (&d)[8]) {
  const uint4 w = *reinterpret_cast<const uint4*>(p);
  const unsigned q[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { d[2 * i] = __uint_as_float(q[i] << 16); d[2 * i + 1] = __uint_as_float(q[i] & 0xFFFF0000u); }
}
__device__ __forceinline__ unsigned bf16_pack2(float a, float b) {     // round-to-nearest-even (v_cvt_pk_bf16_f32)
  bf16x2_t p; p[0] = (__bf16)a; p[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void bf16x8_store(uint16_t* p, const float (&d)[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(bf16_pack2(d[0], d[1]), bf16_pack2(d[2], d[3]), bf16_pack2(d[4], d[5]), bf16_pack2(d[6], d[7]));
}

template <int NV>
__global__ __launch_bounds__(kStyleThreads) void moments_partial_bf16_kernel(const uint16_t* __restrict__ x, PlanePartial* __restrict__ part, int HW, int chunk, int S) {
  __shared__ float red[16];
  const int p = blockIdx.y, s = blockIdx.x;
  const int beg = s * chunk, end = min(HW, beg + chunk);
  const uint16_t* xp = x + (size_t)p * HW;
  float v[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = beg + (j * kStyleThreads + threadIdx.x) * 8;
    if (i < end) bf16x8_load(xp + i, v[j]);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[j][e];
  }
  const float n = (float)(end - beg);
  const float mean = block_sum(sum, red) / n;
  float m2 = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = beg + (j * kStyleThreads + threadIdx.x) * 8;
    if (i < end) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mean; m2 += d * d; }
    }
  }
  m2 = block_sum(m2, red);
  if (threadIdx.x == 0) part[(size_t)p * S + s] = PlanePartial{n, mean, m2, 0.f};
}

__global__ __launch_bounds__(kStyleThreads) void restyle_bf16_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, const float* __restrict__ mu,
                                                                    const float* __restrict__ sig, const float* __restrict__ coefA,
                                                                    const float* __restrict__ coefS, int HW, int chunk) {
  const int p = blockIdx.y;
  const float m = mu[p], a = coefA[p] / sig[p], sh = coefS[p];
  const int beg = blockIdx.x * chunk, end = min(HW, beg + chunk);
  for (int i = beg + threadIdx.x * 8; i < end; i += kStyleThreads * 8) {
    float t[8];
    bf16x8_load(x + (size_t)p * HW + i, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = a * (t[e] - m) + sh;
    bf16x8_store(y + (size_t)p * HW + i, t);
  }
}

// bn_u != nullptr: the fused output-activation backward of the block below, exactly as in restyle_bwd_kernel (ms_style_bwd_actbwd_bf16)
__global__ __launch_bounds__(kStyleThreads) void restyle_bwd_bf16_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x, uint16_t* __restrict__ dx,
                                                                        const float* __restrict__ mu, const float* __restrict__ sig,
                                                                        const float* __restrict__ coefA, float2* __restrict__ part, int HW, int chunk, int S,
                                                                        const uint16_t* __restrict__ bn_u, const float4* __restrict__ bn_coef, float2* __restrict__ bn_part,
                                                                        int C, float slope) {
  __shared__ float red[16];
  const int p = blockIdx.y;
  const float m = mu[p], inv = 1.f / sig[p], a = coefA[p] * inv;
  const int beg = blockIdx.x * chunk, end = min(HW, beg + chunk);
  const float bmean = bn_u ? bn_coef[p % C].z : 0.f;
  float s1 = 0.f, s2 = 0.f, b1 = 0.f, b2 = 0.f;
  for (int i = beg + threadIdx.x * 8; i < end; i += kStyleThreads * 8) {
    float g[8], t[8];
    bf16x8_load(dy + (size_t)p * HW + i, g);
    bf16x8_load(x + (size_t)p * HW + i, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1 += g[e]; s2 += g[e] * ((t[e] - m) * inv); }
    if (dx) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] *= a;
      if (bn_u) {
        float uu[8];
        bf16x8_load(bn_u + (size_t)p * HW + i, uu);
#pragma unroll
        for (int e = 0; e < 8; ++e) { g[e] *= (t[e] > 0.f) ? 1.f : slope; b1 += g[e]; b2 += g[e] * (uu[e] - bmean); }
      }
      bf16x8_store(dx + (size_t)p * HW + i, g);
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) part[(size_t)p * S + blockIdx.x] = make_float2(s1, s2);
  if (bn_u) {
    b1 = block_sum(b1, red);
    b2 = block_sum(b2, red);
    const int c = p % C, n = p / C, N = (int)gridDim.y / C;
    if (threadIdx.x == 0) bn_part[(size_t)c * (N * S) + n * S + blockIdx.x] = make_float2(b1, b2);      // the layout of ms_act_bwd_reduce's partials
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int n,
                            float lr, float b1, float b2, float eps, int step, const int* __restrict__ step_dev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = step_dev ? (*step_dev + 1) : step;
  // torch.optim.Adam (single-tensor path): bias corrections in double on the host there; float pow here
  const double bc1 = 1.0 - pow((double)b1, (double)t);
  const double bc2 = 1.0 - pow((double)b2, (double)t);
  const float gi = g[i];
  const float mi = m[i] * b1 + gi * (1.f - b1);     // m.lerp_(g, 1-b1) == m + (g-m)(1-b1); same to 1 ulp
  const float vi = v[i] * b2 + gi * gi * (1.f - b2);
  m[i] = mi; v[i] = vi;
  const float step_size = (float)((double)lr / bc1);
  const float denom = sqrtf(vi) / (float)sqrt(bc2) + eps;
  p[i] = p[i] - step_size * (mi / denom);
}

__global__ void incr_kernel(int* c) { if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1; }

// ---- the tail of an inner step as ONE launch (ms_step_tail) ---------------------------------------------------------------------------------------
// After the backward pass a step used to end with six tiny dependent launches - style_bwd_finalize per inserted layer, ce_finalize, adam, incr - each ~4.8 us
// of launch boundary around a few KB of work.  The gradient of sample b of a layer is produced entirely by the block (layer, b) of the finalize kernel, and Adam
// on those parameters needs nothing else: so block (b, layer) finalises its row AND takes its Adam step; block (0, 0) also sums the cross-entropy partials of
// the head kernel; the last block to arrive advances the step counter (every block has read it by then - an arrival counter, no spinning).  Same arithmetic,
// same order as the kernels it replaces: the gradients, parameters and the loss are bit-identical.
constexpr int kMaxTailLayers = MS_MAX_TAIL_LAYERS;
struct TailArgs {
  ms_tail_layer layer[kMaxTailLayers];
  const double* ce_part; int ce_nparts; double ce_scale; float* loss_out;
  float* p; float* g; float* m; float* v;
  float lr, b1, b2, eps;
  int* step_dev; int* arrive; int total_blocks, n_layers;
};
__global__ __launch_bounds__(256) void step_tail_kernel(const TailArgs a) {
  __shared__ double redd[16];
  __shared__ float2 stage[1024];
  // (every global load below misses the caches - the data was just written by kernels on other XCDs - so a block is a chain of ~2 us round trips: the
  //  independent ones are issued together up front, and the cross-entropy sum has its own block row instead of queueing behind a layer's work)
  const int slot = *a.step_dev;                    // every block reads the counter before it arrives below
  const int t = slot + 1;
  if ((int)blockIdx.y == a.n_layers) {             // the extra block row: ce_finalize_kernel, restated (same order of additions)
    if (blockIdx.x == 0 && a.ce_part != nullptr) {
      double s = 0.0;
      for (int i = threadIdx.x; i < a.ce_nparts; i += 256) s += a.ce_part[i];
      s = block_sum_d(s, redd);
      if (threadIdx.x == 0) a.loss_out[slot] = (float)(s * a.ce_scale);
    }
  } else {
    const ms_tail_layer& L = a.layer[blockIdx.y];
    const int b = blockIdx.x;
    if (b < L.B) {
      const float2* part = reinterpret_cast<const float2*>(L.part);
      const int C = L.C, S = L.S;
      const int pb = L.perm ? (int)L.perm[b] : b;
      const float lm = (L.off_lmda >= 0) ? a.p[L.off_lmda + b] : 0.f;
      const double bc1 = 1.0 - pow((double)a.b1, (double)t), bc2 = 1.0 - pow((double)a.b2, (double)t);
      const float step_size = (float)((double)a.lr / bc1), rs2 = (float)sqrt(bc2);
      auto adam = [&](int i, float gi) {             // adam_kernel's arithmetic, bias corrections hoisted
        const float mi = a.m[i] * a.b1 + gi * (1.f - a.b1);
        const float vi = a.v[i] * a.b2 + gi * gi * (1.f - a.b2);
        a.m[i] = mi; a.v[i] = vi;
        const float denom = sqrtf(vi) / rs2 + a.eps;
        a.p[i] = a.p[i] - step_size * (mi / denom);
      };
      double acc = 0.0;
      // the S partial records of the block's C channels: one load per thread where they fit (a 16-channel layer kept 16 threads busy with S dependent ~2 us
      // round trips each), parked in LDS and summed per channel in the order s = 0, 1, ... as before (the same bits)
      const bool staged = (C * S <= 1024);
      if (staged) {
        for (int i = threadIdx.x; i < C * S; i += blockDim.x) stage[i] = part[(size_t)b * C * S + i];
        __syncthreads();
      }
      for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int p = b * C + c;
        const int q = pb * C + c;
        const double dsig = (double)L.sig[q] - (double)L.sig[p], dmu = (double)L.mu[q] - (double)L.mu[p];
        double s1 = 0.0, s2 = 0.0;
        if (staged) { for (int s = 0; s < S; ++s) { const float2 v2 = stage[c * S + s]; s1 += (double)v2.x; s2 += (double)v2.y; } }
        else { for (int s = 0; s < S; ++s) { const float2 v2 = part[(size_t)p * S + s]; s1 += (double)v2.x; s2 += (double)v2.y; } }
        if (L.off_gamma >= 0) {
          const float dg = (float)((double)L.gamma_std[c] * s2), db = (float)((double)L.beta_std[c] * s1);
          a.g[L.off_gamma + p] = dg; a.g[L.off_beta + p] = db;
          if (L.learn_noise) { adam(L.off_gamma + p, dg); adam(L.off_beta + p, db); }
        }
        acc += dsig * s2 + dmu * s1;
      }
      acc = block_sum_d(acc, redd);
      if (threadIdx.x == 0 && L.off_lmda >= 0) {
        const float dl = (lm >= 0.f && lm <= 1.f) ? (float)acc : 0.f;       // clamp(): zero gradient outside [0,1]
        a.g[L.off_lmda + b] = dl;
        if (L.learn_mix) adam(L.off_lmda + b, dl);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(a.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == a.total_blocks - 1) {
      __hip_atomic_store(a.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
      *a.step_dev = t;
    }
  }
}

struct Split { int chunk, S, nv; bool vec; };

static Split choose_split(int P, int HW, bool vec_ok) {
  // Largest register tile that still yields >= ~1024 blocks (256 CUs x 4): P*S >= 1024 when possible.
  Split sp; sp.vec = vec_ok;
  const int vec = vec_ok ? 4 : 1;
  int nv = 8;
  while (nv > 1) {
    const int chunk = kStyleThreads * vec * nv;
    const long blocks = (long)P * cdiv(HW, chunk);
    if (blocks >= 1024) break;
    nv >>= 1;
  }
  sp.nv = nv;
  sp.chunk = kStyleThreads * vec * nv;
  sp.S = cdiv(HW, sp.chunk);
  return sp;
}

}  // namespace ms

using namespace ms;

// workspace layout: [0, scratch) per-launch partials of the three-kernel forward and of the backward | [scratch, +fused) persistent state
// of the single-read kernel (epoch + tagged granules: zero-filled once by the caller, never touched by the other kernels)
static size_t style_scratch_bytes(int B, int C, int HW) {
  // worst case split: scalar path, nv=1 -> chunk 256
  const size_t P = (size_t)B * C;
  const size_t S = (size_t)cdiv(HW, kStyleThreads);
  return (P * S * sizeof(PlanePartial) + 256 + 15) / 16 * 16;
}

extern "C" size_t ms_style_ws_bytes(int B, int C, int HW) {
  return style_scratch_bytes(B, C, HW) + (ms_style_fused_ws_bytes(B, C, HW) + 15) / 16 * 16;
}

// byte offset of the single-read kernel's state block inside a style workspace (its int[1] is the error word), or (size_t)-1 when
// the shape has no single-read path
extern "C" size_t ms_style_ws_state_offset(int B, int C, int HW) {
  return ms_style_fused_ws_bytes(B, C, HW) == 0 ? (size_t)-1 : style_scratch_bytes(B, C, HW);
}

static int launch_moments(const float* x, PlanePartial* part, int P, int HW, const Split& sp, hipStream_t st) {
  dim3 grid(sp.S, P), block(kStyleThreads);
#define MS_LAUNCH_MOM(V, N) MS_LAUNCH((moments_partial_kernel<V, N>), grid, block, 0, st, x, part, HW, sp.chunk, sp.S)
  if (sp.vec) {
    switch (sp.nv) { case 8: MS_LAUNCH_MOM(4, 8); break; case 4: MS_LAUNCH_MOM(4, 4); break; case 2: MS_LAUNCH_MOM(4, 2); break; default: MS_LAUNCH_MOM(4, 1); }
  } else {
    switch (sp.nv) { case 8: MS_LAUNCH_MOM(1, 8); break; case 4: MS_LAUNCH_MOM(1, 4); break; case 2: MS_LAUNCH_MOM(1, 2); break; default: MS_LAUNCH_MOM(1, 1); }
  }
#undef MS_LAUNCH_MOM
  return check_launch("moments_partial");
}

static int check_style_args(int B, int C, int HW, const void* ws, size_t ws_bytes) {
  if (B < 1 || C < 1 || HW < 2) { set_error("ms_style: invalid shape B=%d C=%d HW=%d", B, C, HW); return MS_ERR_INVALID; }
  if ((long)B * C > 2147483647L / 2 || (long)B * C > 65535L * 65535L) { set_error("ms_style: too many planes"); return MS_ERR_INVALID; }
  if (ws == nullptr || ws_bytes < ms_style_ws_bytes(B, C, HW)) { set_error("ms_style: workspace too small (%zu < %zu)", ws_bytes, ms_style_ws_bytes(B, C, HW)); return MS_ERR_WORKSPACE; }
  if (!aligned16(ws)) { set_error("ms_style: workspace must be 16-byte aligned"); return MS_ERR_ALIGN; }
  return MS_OK;
}

// gridDim.y is limited to 65535: planes beyond that are processed in slabs.
static constexpr int kMaxPlanesPerLaunch = 65535;

extern "C" int ms_style_moments(const float* x, float* mu, float* sig, int planes, int HW, float eps, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_style_args(planes, 1, HW, ws, ws_bytes)) return e;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (HW % 4 == 0) && aligned16(x);
  const Split sp = choose_split(planes, HW, vec);
  PlanePartial* part = (PlanePartial*)ws;
  for (int p0 = 0; p0 < planes; p0 += kMaxPlanesPerLaunch) {
    const int np = std::min(kMaxPlanesPerLaunch, planes - p0);
    if (int e = launch_moments(x + (size_t)p0 * HW, part + (size_t)p0 * sp.S, np, HW, sp, st)) return e;
  }
  MS_LAUNCH(moments_merge_kernel, dim3(cdiv(planes, 256)), dim3(256), 0, st, part, mu, sig, planes, sp.S, HW, eps);
  return check_launch("moments_merge");
}

extern "C" int ms_style_coeffs(float* mu, float* sig, float* gamma_std, float* beta_std, int compute_std, const float* lmda,
                               const float* gamma_noise, const float* beta_noise, const int64_t* perm, float* coefA, float* coefS,
                               int B, int C, void* stream) {
  if (B < 1 || C < 1) { set_error("ms_style_coeffs: invalid shape"); return MS_ERR_INVALID; }
  if ((compute_std & 1) && B < 2) { set_error("ms_style_coeffs: batch std needs B >= 2"); return MS_ERR_INVALID; }
  if (lmda != nullptr && perm == nullptr) { set_error("ms_style_coeffs: mixing needs perm"); return MS_ERR_INVALID; }
  if ((gamma_noise == nullptr) != (beta_noise == nullptr)) { set_error("ms_style_coeffs: gamma/beta noise must both be given"); return MS_ERR_INVALID; }
  MS_LAUNCH(style_finalize_kernel, dim3(C), dim3(256), 2 * B * sizeof(float), (hipStream_t)stream, (const PlanePartial*)nullptr, mu, sig,
                     gamma_std, beta_std, compute_std, lmda, gamma_noise, beta_noise, perm, coefA, coefS, B, C, 0, 0, 0.f);
  return check_launch("style_finalize");
}

extern "C" int ms_style_apply(const float* x, float* y, const float* mu, const float* sig, const float* coefA, const float* coefS,
                              int planes, int HW, void* stream) {
  if (planes < 1 || HW < 1) { set_error("ms_style_apply: invalid shape"); return MS_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (HW % 4 == 0) && aligned16(x) && aligned16(y);
  const Split sp = choose_split(planes, HW, vec);
  for (int p0 = 0; p0 < planes; p0 += kMaxPlanesPerLaunch) {
    const int np = std::min(kMaxPlanesPerLaunch, planes - p0);
    dim3 grid(sp.S, np), block(kStyleThreads);
    const size_t off = (size_t)p0 * HW;
    if (vec) MS_LAUNCH(restyle_kernel<4>, grid, block, 0, st, x + off, y + off, mu + p0, sig + p0, coefA + p0, coefS + p0, HW, sp.chunk);
    else MS_LAUNCH(restyle_kernel<1>, grid, block, 0, st, x + off, y + off, mu + p0, sig + p0, coefA + p0, coefS + p0, HW, sp.chunk);
  }
  return check_launch("restyle");
}

extern "C" int ms_style_fwd(const float* x, float* y, float* mu, float* sig, float* gamma_std, float* beta_std, int compute_std,
                            const float* lmda, const float* gamma_noise, const float* beta_noise, const int64_t* perm,
                            float* coefA, float* coefS, int B, int C, int HW, float eps, void* ws, size_t ws_bytes, void* stream) {
  // single-read kernel when the shape is eligible (option "style.fused" = 0 forces the three-kernel path, for A/B timing).  compute_std bit 2
  // (MS_STYLE_SHARED_DEVICE): the caller runs other kernels beside this one (side streams, another process) - the single-read kernel's
  // co-residency argument does not hold then, so the three-launch path is taken.
  const bool fused_on = opt(OPT_STYLE_FUSED) != 0;
  constexpr size_t min_kb = 1024;
  const bool big = (size_t)B * C * HW * sizeof(float) >= (min_kb << 10);
  const bool shared = (compute_std & 4) != 0;
  compute_std &= 3;
  const size_t fb = (fused_on && big && !shared) ? ms_style_fused_ws_bytes(B, C, HW) : 0;
  const size_t off = style_scratch_bytes(B, C, HW);
  if (fb != 0 && ws != nullptr && ws_bytes >= off + fb && aligned16(ws) && aligned16(x) && aligned16(y))
    return ms_style_fwd_fused(x, y, mu, sig, gamma_std, beta_std, compute_std, lmda, gamma_noise, beta_noise, perm, coefA, coefS, B, C, HW, eps,
                              (char*)ws + off, ws_bytes - off, stream);
  return ms_style_fwd_3k(x, y, mu, sig, gamma_std, beta_std, compute_std, lmda, gamma_noise, beta_noise, perm, coefA, coefS, B, C, HW, eps, ws, ws_bytes, stream);
}

extern "C" int ms_style_fwd_3k(const float* x, float* y, float* mu, float* sig, float* gamma_std, float* beta_std, int compute_std,
                               const float* lmda, const float* gamma_noise, const float* beta_noise, const int64_t* perm,
                               float* coefA, float* coefS, int B, int C, int HW, float eps, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_style_args(B, C, HW, ws, ws_bytes)) return e;
  if ((compute_std & 1) && B < 2) { set_error("ms_style_fwd: batch std needs B >= 2"); return MS_ERR_INVALID; }
  if (lmda != nullptr && perm == nullptr) { set_error("ms_style_fwd: mixing needs perm"); return MS_ERR_INVALID; }
  if ((gamma_noise == nullptr) != (beta_noise == nullptr)) { set_error("ms_style_fwd: gamma/beta noise must both be given"); return MS_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const int P = B * C;
  const bool vec = (HW % 4 == 0) && aligned16(x);
  const Split sp = choose_split(P, HW, vec);
  PlanePartial* part = (PlanePartial*)ws;
  for (int p0 = 0; p0 < P; p0 += kMaxPlanesPerLaunch) {
    const int np = std::min(kMaxPlanesPerLaunch, P - p0);
    if (int e = launch_moments(x + (size_t)p0 * HW, part + (size_t)p0 * sp.S, np, HW, sp, st)) return e;
  }
  MS_LAUNCH(style_finalize_kernel, dim3(C), dim3(256), 2 * B * sizeof(float), st, (const PlanePartial*)part, mu, sig, gamma_std, beta_std,
                     compute_std, lmda, gamma_noise, beta_noise, perm, coefA, coefS, B, C, sp.S, HW, eps);
  if (int e = check_launch("style_finalize")) return e;
  if (y == nullptr) return MS_OK;                      // statistics and coefficients only (see ms_style_fwd)
  return ms_style_apply(x, y, mu, sig, coefA, coefS, P, HW, stream);
}

static int style_bwd_impl(const float* dy, const float* x, float* dx, const float* mu, const float* sig, const float* coefA,
                          const float* gamma_std, const float* beta_std, const float* lmda, const int64_t* perm,
                          float* d_gamma, float* d_beta, float* d_lmda, int B, int C, int HW, void* ws, size_t ws_bytes,
                          const float* bn_u, const float* bn_coef4, float* bn_part, float slope, void* stream, const HeadSrc* head = nullptr) {
  if (int e = check_style_args(B, C, HW, ws, ws_bytes)) return e;
  if (d_lmda != nullptr && (lmda == nullptr || perm == nullptr)) { set_error("ms_style_bwd: d_lmda needs lmda and perm"); return MS_ERR_INVALID; }
  const HeadSrc hd = head ? *head : HeadSrc{nullptr, nullptr, nullptr, 0};
  if (head != nullptr && (hd.g == nullptr || hd.w == nullptr || hd.K < 1 || hd.K > 4 || HW % 4 != 0 || !aligned16(hd.g) || (hd.out != nullptr && !aligned16(hd.out)) || !aligned16(x))) {
    set_error("ms_style_bwd_head: head gradient [N,K,HW] (K <= 4), weights [K][C], H*W %% 4 == 0 and 16-byte aligned tensors are required"); return MS_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const int P = B * C;
  if (bn_u != nullptr && (dx == nullptr || bn_coef4 == nullptr || bn_part == nullptr || P > kMaxPlanesPerLaunch || !aligned16(bn_coef4))) {
    set_error("ms_style_bwd_actbwd: needs dx, bn_coef4 (16-byte aligned), bn_part and <= 65535 planes"); return MS_ERR_INVALID;
  }
  const bool vec = (HW % 4 == 0) && aligned16(x) && (head != nullptr || aligned16(dy)) && (dx == nullptr || aligned16(dx)) && (bn_u == nullptr || aligned16(bn_u));
  if (head != nullptr && !vec) { set_error("ms_style_bwd_head: the vector path is required (16-byte aligned x, dx, bn_u)"); return MS_ERR_ALIGN; }
  const Split sp = choose_split(P, HW, vec);
  float2* part = (float2*)ws;
  for (int p0 = 0; p0 < P; p0 += kMaxPlanesPerLaunch) {
    const int np = std::min(kMaxPlanesPerLaunch, P - p0);
    dim3 grid(sp.S, np), block(kStyleThreads);
    const size_t off = (size_t)p0 * HW;
    float* dxo = dx ? dx + off : nullptr;
    if (head != nullptr && p0 != 0) { set_error("ms_style_bwd_head: more than 65535 planes"); return MS_ERR_INVALID; }
    if (vec) MS_LAUNCH(restyle_bwd_kernel<4>, grid, block, 0, st, dy ? dy + off : nullptr, x + off, dxo, mu + p0, sig + p0, coefA + p0, part + (size_t)p0 * sp.S, HW, sp.chunk, sp.S,
                       bn_u, (const float4*)bn_coef4, (float2*)bn_part, C, slope, hd);
    else MS_LAUNCH(restyle_bwd_kernel<1>, grid, block, 0, st, dy + off, x + off, dxo, mu + p0, sig + p0, coefA + p0, part + (size_t)p0 * sp.S, HW, sp.chunk, sp.S,
                   bn_u, (const float4*)bn_coef4, (float2*)bn_part, C, slope, hd);
  }
  if (int e = check_launch("restyle_bwd")) return e;
  if (d_gamma || d_beta || d_lmda) {
    MS_LAUNCH(style_bwd_finalize_kernel, dim3(B), dim3(256), 0, st, (const float2*)part, mu, sig, gamma_std, beta_std, lmda, perm,
                       d_gamma, d_beta, d_lmda, B, C, sp.S);
    return check_launch("style_bwd_finalize");
  }
  return MS_OK;
}

extern "C" int ms_style_bwd(const float* dy, const float* x, float* dx, const float* mu, const float* sig, const float* coefA,
                            const float* gamma_std, const float* beta_std, const float* lmda, const int64_t* perm,
                            float* d_gamma, float* d_beta, float* d_lmda, int B, int C, int HW, void* ws, size_t ws_bytes, void* stream) {
  return style_bwd_impl(dy, x, dx, mu, sig, coefA, gamma_std, beta_std, lmda, perm, d_gamma, d_beta, d_lmda, B, C, HW, ws, ws_bytes, nullptr, nullptr, nullptr, 1.f, stream);
}

extern "C" int ms_style_bwd_actbwd_parts(int B, int C, int HW) {
  return B * choose_split(B * C, HW, HW % 4 == 0).S;
}

extern "C" int ms_style_bwd_actbwd(const float* dy, const float* x, float* dx, const float* mu, const float* sig, const float* coefA,
                                   const float* gamma_std, const float* beta_std, const float* lmda, const int64_t* perm,
                                   float* d_gamma, float* d_beta, float* d_lmda, int B, int C, int HW, void* ws, size_t ws_bytes,
                                   const float* bn_u, const float* bn_coef4, float* bn_part, float act_slope, void* stream) {
  if (bn_u == nullptr) { set_error("ms_style_bwd_actbwd: bn_u is required"); return MS_ERR_INVALID; }
  if (HW % 4 != 0) { set_error("ms_style_bwd_actbwd: H*W must be a multiple of 4 (the partial count is fixed by ms_style_bwd_actbwd_parts)"); return MS_ERR_INVALID; }
  return style_bwd_impl(dy, x, dx, mu, sig, coefA, gamma_std, beta_std, lmda, perm, d_gamma, d_beta, d_lmda, B, C, HW, ws, ws_bytes, bn_u, bn_coef4, bn_part, act_slope, stream);
}

// ms_style_bwd / ms_style_bwd_actbwd of a layer that sits directly in front of a 1x1 head (+ sigmoid): the layer's incoming gradient dy = ms_head_bwd(g, out, w)
// is never written - this launch forms it from g [B,K,HW], out [B,K,HW] (the head's sigmoid output; NULL: no sigmoid) and w [K][C] while it streams.
// bn_u / bn_coef4 / bn_part may be NULL (no activation backward); dx may be NULL.
extern "C" int ms_style_bwd_head(const float* head_g, const float* head_out, const float* head_w, int K, const float* x, float* dx, const float* mu, const float* sig,
                                 const float* coefA, const float* gamma_std, const float* beta_std, const float* lmda, const int64_t* perm,
                                 float* d_gamma, float* d_beta, float* d_lmda, int B, int C, int HW, void* ws, size_t ws_bytes,
                                 const float* bn_u, const float* bn_coef4, float* bn_part, float act_slope, void* stream) {
  const HeadSrc hd{head_g, head_out, head_w, K};
  return style_bwd_impl(nullptr, x, dx, mu, sig, coefA, gamma_std, beta_std, lmda, perm, d_gamma, d_beta, d_lmda, B, C, HW, ws, ws_bytes, bn_u, bn_coef4, bn_part, act_slope, stream, &hd);
}

// ---- bf16 activation storage -----------------------------------------------------------------------------------------------
static Split choose_split_bf16(int P, int HW) {
  Split sp; sp.vec = true;
  int nv = 8;                                          // 8 values per access: <= 64 fp32 registers of data per thread
  while (nv > 1) { if ((long)P * cdiv(HW, kStyleThreads * 8 * nv) >= 1024) break; nv >>= 1; }
  sp.nv = nv; sp.chunk = kStyleThreads * 8 * nv; sp.S = cdiv(HW, sp.chunk);
  return sp;
}

static int check_bf16_args(const char* who, const void* a, const void* b, const void* c, int HW) {
  if (HW % 8 != 0) { set_error("%s: bf16 storage needs H*W %% 8 == 0 (16-byte accesses)", who); return MS_ERR_INVALID; }
  if (!aligned16(a) || !aligned16(b) || (c != nullptr && !aligned16(c))) { set_error("%s: bf16 tensors must be 16-byte aligned", who); return MS_ERR_ALIGN; }
  return MS_OK;
}

extern "C" size_t ms_style_ws_bytes_bf16(int B, int C, int HW) {
  return style_scratch_bytes(B, C, HW) + (ms_style_fused_ws_bytes_bf16(B, C, HW) + 15) / 16 * 16;
}

extern "C" int ms_style_fwd_bf16(const uint16_t* x, uint16_t* y, float* mu, float* sig, float* gamma_std, float* beta_std, int compute_std,
                                 const float* lmda, const float* gamma_noise, const float* beta_noise, const int64_t* perm,
                                 float* coefA, float* coefS, int B, int C, int HW, float eps, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_bf16_args("ms_style_fwd_bf16", x, y, nullptr, HW)) return e;
  if (B < 1 || C < 1 || ws == nullptr || !aligned16(ws) || ws_bytes < ms_style_ws_bytes_bf16(B, C, HW)) { set_error("ms_style_fwd_bf16: invalid shape or workspace"); return MS_ERR_WORKSPACE; }
  const bool shared = (compute_std & 4) != 0;
  compute_std &= 3;
  const size_t fb = shared ? 0 : ms_style_fused_ws_bytes_bf16(B, C, HW);
  const size_t off = style_scratch_bytes(B, C, HW);
  if (fb != 0 && (size_t)B * C * HW * 2 >= (1u << 20))
    return ms_style_fwd_fused_bf16(x, y, mu, sig, gamma_std, beta_std, compute_std, lmda, gamma_noise, beta_noise, perm, coefA, coefS, B, C, HW, eps,
                                   (char*)ws + off, ws_bytes - off, stream);
  if ((compute_std & 1) && B < 2) { set_error("ms_style_fwd_bf16: batch std needs B >= 2"); return MS_ERR_INVALID; }
  if (lmda != nullptr && perm == nullptr) { set_error("ms_style_fwd_bf16: mixing needs perm"); return MS_ERR_INVALID; }
  if ((gamma_noise == nullptr) != (beta_noise == nullptr)) { set_error("ms_style_fwd_bf16: gamma/beta noise must both be given"); return MS_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const int P = B * C;
  if (P > kMaxPlanesPerLaunch) { set_error("ms_style_fwd_bf16: too many planes"); return MS_ERR_INVALID; }
  const Split sp = choose_split_bf16(P, HW);
  PlanePartial* part = (PlanePartial*)ws;
  dim3 grid(sp.S, P), block(kStyleThreads);
  switch (sp.nv) {
    case 8: MS_LAUNCH(moments_partial_bf16_kernel<8>, grid, block, 0, st, x, part, HW, sp.chunk, sp.S); break;
    case 4: MS_LAUNCH(moments_partial_bf16_kernel<4>, grid, block, 0, st, x, part, HW, sp.chunk, sp.S); break;
    case 2: MS_LAUNCH(moments_partial_bf16_kernel<2>, grid, block, 0, st, x, part, HW, sp.chunk, sp.S); break;
    default: MS_LAUNCH(moments_partial_bf16_kernel<1>, grid, block, 0, st, x, part, HW, sp.chunk, sp.S);
  }
  if (int e = check_launch("moments_partial_bf16")) return e;
  MS_LAUNCH(style_finalize_kernel, dim3(C), dim3(256), 2 * B * sizeof(float), st, (const PlanePartial*)part, mu, sig, gamma_std, beta_std,
                     compute_std, lmda, gamma_noise, beta_noise, perm, coefA, coefS, B, C, sp.S, HW, eps);
  if (int e = check_launch("style_finalize")) return e;
  if (y == nullptr) return MS_OK;
  MS_LAUNCH(restyle_bf16_kernel, grid, block, 0, st, x, y, (const float*)mu, (const float*)sig, (const float*)coefA, (const float*)coefS, HW, sp.chunk);
  return check_launch("restyle_bf16");
}

static int style_bwd_bf16_impl(const uint16_t* dy, const uint16_t* x, uint16_t* dx, const float* mu, const float* sig, const float* coefA,
                               const float* gamma_std, const float* beta_std, const float* lmda, const int64_t* perm,
                               float* d_gamma, float* d_beta, float* d_lmda, int B, int C, int HW, void* ws, size_t ws_bytes,
                               const uint16_t* bn_u, const float* bn_coef4, float* bn_part, float slope, void* stream) {
  if (int e = check_bf16_args("ms_style_bwd_bf16", dy, x, dx, HW)) return e;
  if (bn_u != nullptr && (dx == nullptr || bn_coef4 == nullptr || bn_part == nullptr || !aligned16(bn_u) || !aligned16(bn_coef4))) {
    set_error("ms_style_bwd_actbwd_bf16: needs dx, bn_u and bn_coef4 (16-byte aligned) and bn_part"); return MS_ERR_INVALID;
  }
  if (int e = check_style_args(B, C, HW, ws, ws_bytes)) return e;
  if (d_lmda != nullptr && (lmda == nullptr || perm == nullptr)) { set_error("ms_style_bwd_bf16: d_lmda needs lmda and perm"); return MS_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const int P = B * C;
  if (P > kMaxPlanesPerLaunch) { set_error("ms_style_bwd_bf16: too many planes"); return MS_ERR_INVALID; }
  const Split sp = choose_split_bf16(P, HW);
  float2* part = (float2*)ws;
  MS_LAUNCH(restyle_bwd_bf16_kernel, dim3(sp.S, P), dim3(kStyleThreads), 0, st, dy, x, dx, mu, sig, coefA, part, HW, sp.chunk, sp.S,
            bn_u, (const float4*)bn_coef4, (float2*)bn_part, C, slope);
  if (int e = check_launch("restyle_bwd_bf16")) return e;
  if (d_gamma || d_beta || d_lmda) {
    MS_LAUNCH(style_bwd_finalize_kernel, dim3(B), dim3(256), 0, st, (const float2*)part, mu, sig, gamma_std, beta_std, lmda, perm, d_gamma, d_beta, d_lmda, B, C, sp.S);
    return check_launch("style_bwd_finalize");
  }
  return MS_OK;
}
extern "C" int ms_style_bwd_bf16(const uint16_t* dy, const uint16_t* x, uint16_t* dx, const float* mu, const float* sig, const float* coefA,
                                 const float* gamma_std, const float* beta_std, const float* lmda, const int64_t* perm,
                                 float* d_gamma, float* d_beta, float* d_lmda, int B, int C, int HW, void* ws, size_t ws_bytes, void* stream) {
  return style_bwd_bf16_impl(dy, x, dx, mu, sig, coefA, gamma_std, beta_std, lmda, perm, d_gamma, d_beta, d_lmda, B, C, HW, ws, ws_bytes, nullptr, nullptr, nullptr, 0.f, stream);
}
// bf16 twin of ms_style_bwd_actbwd; bn_part holds [C][ms_style_bwd_actbwd_parts_bf16(B, C, HW)] float2 partials
extern "C" int ms_style_bwd_actbwd_parts_bf16(int B, int C, int HW) { return B * choose_split_bf16(B * C, HW).S; }
extern "C" int ms_style_bwd_actbwd_bf16(const uint16_t* dy, const uint16_t* x, uint16_t* dx, const float* mu, const float* sig, const float* coefA,
                                        const float* gamma_std, const float* beta_std, const float* lmda, const int64_t* perm,
                                        float* d_gamma, float* d_beta, float* d_lmda, int B, int C, int HW, void* ws, size_t ws_bytes,
                                        const uint16_t* bn_u, const float* bn_coef4, float* bn_part, float act_slope, void* stream) {
  if (bn_u == nullptr) { set_error("ms_style_bwd_actbwd_bf16: bn_u is required"); return MS_ERR_INVALID; }
  return style_bwd_bf16_impl(dy, x, dx, mu, sig, coefA, gamma_std, beta_std, lmda, perm, d_gamma, d_beta, d_lmda, B, C, HW, ws, ws_bytes, bn_u, bn_coef4, bn_part, act_slope, stream);
}

extern "C" int ms_adam_step(float* p, const float* g, float* m, float* v, int n, float lr, float b1, float b2, float eps, int step,
                            const int* step_dev, void* stream) {
  if (n < 0 || (step_dev == nullptr && step < 1)) { set_error("ms_adam_step: invalid n/step"); return MS_ERR_INVALID; }
  if (n == 0) return MS_OK;
  MS_LAUNCH(adam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2, eps, step, step_dev);
  return check_launch("adam");
}

extern "C" int ms_style_bwd_slots(int B, int C, int HW, int bf16) {
  return bf16 ? choose_split_bf16(B * C, HW).S : choose_split(B * C, HW, HW % 4 == 0).S;
}

extern "C" int ms_step_tail(const ms_tail_layer* layers, int n_layers, const double* ce_part, int ce_nparts, double ce_scale, float* loss_out,
                            float* p, float* g, float* m, float* v, float lr, float b1, float b2, float eps, int* step_dev, int* arrive, void* stream) {
  if ((layers == nullptr && n_layers > 0) || n_layers < 0 || n_layers > kMaxTailLayers || p == nullptr || g == nullptr || m == nullptr || v == nullptr || step_dev == nullptr ||
      arrive == nullptr || (ce_part != nullptr && (loss_out == nullptr || ce_nparts < 1))) {
    set_error("ms_step_tail: 0..%d layers, flat p/g/m/v, step_dev, arrive (zero-initialised, dedicated) are required", kMaxTailLayers); return MS_ERR_INVALID;
  }
  TailArgs a;
  int maxB = 1;
  for (int i = 0; i < n_layers; ++i) {
    const ms_tail_layer& L = layers[i];
    if (L.part == nullptr || L.mu == nullptr || L.sig == nullptr || L.B < 1 || L.C < 1 || L.S < 1 || (L.off_gamma >= 0 && (L.gamma_std == nullptr || L.beta_std == nullptr || L.off_beta < 0)) ||
        (L.off_lmda >= 0 && L.perm == nullptr)) {
      set_error("ms_step_tail: layer %d is incomplete", i); return MS_ERR_INVALID;
    }
    a.layer[i] = L;
    maxB = std::max(maxB, L.B);
  }
  a.ce_part = ce_part; a.ce_nparts = ce_nparts; a.ce_scale = ce_scale; a.loss_out = loss_out;
  a.p = p; a.g = g; a.m = m; a.v = v; a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps;
  const int gy = n_layers + 1;                       // + the cross-entropy block row
  a.step_dev = step_dev; a.arrive = arrive; a.total_blocks = maxB * gy; a.n_layers = n_layers;
  MS_LAUNCH(step_tail_kernel, dim3(maxB, gy), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("step_tail");
}

extern "C" int ms_counter_incr(int* counter, void* stream) {
  MS_LAUNCH(incr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter);
  return check_launch("incr");
}
