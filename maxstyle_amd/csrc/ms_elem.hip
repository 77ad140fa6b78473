// Streaming (HBM-bound) kernels around the convolutions: BatchNorm apply + residual + LeakyReLU, the backward
// mask+reduction passes of BatchNorm in "batch statistics, frozen affine" mode, 2x2 sum pooling (gradient of nearest
// up-sampling), and the two network heads (1x1 conv + sigmoid; 1x1 conv + log-softmax + NLL with fused backward).
//
// Reference ops replaced: nn.BatchNorm2d (train mode, track_running_stats=False; model_util.py:468-510),
// nn.LeakyReLU(0.2)/nn.ReLU, residual add (encoder_decoder.py:62-64, 344-346), nn.UpsamplingNearest2d backward,
// MyDecoder.final_conv + nn.Sigmoid (encoder_decoder.py:582,594), cross_entropy_2D (custom_loss.py:1043-1078).
// All reductions are fixed-order two-stage (per-block partials, fp64 finalize): deterministic, no float atomics.
#include <algorithm>
#include "ms_common.h"
#include "maxstyle_hip.h"

namespace ms {

constexpr int kElemThreads = 256;

struct ElemSplit { int chunk, S; };
static ElemSplit elem_split(int planes, int HW) {
  // ~2048 blocks target, chunk multiple of 1024 elements (256 threads x float4)
  int chunk = 1024;
  while ((long)planes * cdiv(HW, chunk) > 4096 && chunk < 16384) chunk <<= 1;
  return ElemSplit{chunk, cdiv(HW, chunk)};
}

// out = LeakyReLU_slope(sc[c]*u + sh[c] + res)   res: none | same shape | half resolution (nearest up-sampled on the fly)
template <int RES /*0 none,1 same,2 half-res*/, typename AT = float>
__global__ __launch_bounds__(kElemThreads) void bn_act_kernel(const void* __restrict__ u, const float4* __restrict__ coef, const void* __restrict__ res,
                                                              void* __restrict__ out, int C, int H, int W, int chunk, float slope) {
  using IO = ActIO<AT>;
  const int p = blockIdx.y, c = p % C;
  const float4 cf = coef[c];
  const int HW = H * W;
  const int beg = blockIdx.x * chunk, end = min(HW, beg + chunk);
  const size_t pb = (size_t)p * HW;
  const bool vec = (W % 4 == 0);
  if (vec) {
    const size_t rb = (RES == 1) ? pb : (RES == 2 ? (size_t)p * (HW / 4) : 0);
    for (int i = beg + threadIdx.x * 4; i < end; i += kElemThreads * 4) {
      float4 t = IO::ld4(u, pb + i);
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (RES == 1) r = IO::ld4(res, rb + i);
      if (RES == 2) {
        const int y = i / W, x = i - y * W;            // x % 4 == 0
        const float2 h = IO::ld2(res, rb + (size_t)(y >> 1) * (W >> 1) + (x >> 1));
        r = make_float4(h.x, h.x, h.y, h.y);
      }
      t.x = leaky(cf.x * t.x + cf.y + r.x, slope); t.y = leaky(cf.x * t.y + cf.y + r.y, slope);
      t.z = leaky(cf.x * t.z + cf.y + r.z, slope); t.w = leaky(cf.x * t.w + cf.y + r.w, slope);
      IO::st4(out, pb + i, t);
    }
  } else {
    for (int i = beg + threadIdx.x; i < end; i += kElemThreads) {
      float r = 0.f;
      if (RES == 1) r = IO::ld1(res, pb + i);
      if (RES == 2) { const int y = i / W, x = i - y * W; r = IO::ld1(res, (size_t)p * ((H / 2) * (W / 2)) + (size_t)(y >> 1) * (W >> 1) + (x >> 1)); }
      IO::st1(out, pb + i, leaky(cf.x * IO::ld1(u, pb + i) + cf.y + r, slope));
    }
  }
}

// g = gin * (ref > 0 ? 1 : slope); partial sums of g and g*(u - mean_c) per (plane, chunk) -> part[c][n*S+s].
// The second sum is CENTRED with the channel mean of the forward pass (coef.z): sum g*u - mean*sum g cancels catastrophically when the conv
// output has a large mean (a 1e-7 perturbation of `mean` moved weight gradients by up to 0.7 %); nn.BatchNorm2d's backward centres too.
//   MASK 0: ref = the materialised activation output (sign(out) == sign(pre-activation))
//   MASK 1: ref = sc[c]*u + sh[c]  (activation whose output was never materialised: folded into the next conv's prologue)
typedef unsigned long long u64;

template <int MASK, typename AT = float>
__global__ __launch_bounds__(kElemThreads) void act_bwd_reduce_kernel(const void* __restrict__ gin, const void* __restrict__ ref, const void* __restrict__ u,
                                                                      const float4* __restrict__ coef, void* __restrict__ gout, float2* __restrict__ part,
                                                                      int C, int HW, int chunk, int S, int N, float slope) {
  __shared__ float red[16];
  const int p = blockIdx.y, c = p % C, n = p / C;
  const float4 cf = coef[c];
  const int beg = blockIdx.x * chunk, end = min(HW, beg + chunk);
  const size_t base = (size_t)p * HW;
  using IO = ActIO<AT>;
  float s1 = 0.f, s2 = 0.f;
  if (HW % 4 == 0) {
    for (int i = beg + threadIdx.x * 4; i < end; i += kElemThreads * 4) {
      const float4 g = IO::ld4(gin, base + i);
      const float4 uu = IO::ld4(u, base + i);
      float4 r;
      if (MASK == 0) r = IO::ld4(ref, base + i);
      else r = make_float4(cf.x * uu.x + cf.y, cf.x * uu.y + cf.y, cf.x * uu.z + cf.y, cf.x * uu.w + cf.y);
      float4 o;
      o.x = g.x * (r.x > 0.f ? 1.f : slope); o.y = g.y * (r.y > 0.f ? 1.f : slope);
      o.z = g.z * (r.z > 0.f ? 1.f : slope); o.w = g.w * (r.w > 0.f ? 1.f : slope);
      IO::st4(gout, base + i, o);
      s1 += (o.x + o.y) + (o.z + o.w);
      s2 += (o.x * (uu.x - cf.z) + o.y * (uu.y - cf.z)) + (o.z * (uu.z - cf.z) + o.w * (uu.w - cf.z));     // centred: sum g*(u - mean)
    }
  } else {
    for (int i = beg + threadIdx.x; i < end; i += kElemThreads) {
      const float g = IO::ld1(gin, base + i), uu = IO::ld1(u, base + i);
      const float r = (MASK == 0) ? IO::ld1(ref, base + i) : (cf.x * uu + cf.y);
      const float o = g * (r > 0.f ? 1.f : slope);
      IO::st1(gout, base + i, o);
      s1 += o; s2 += o * (uu - cf.z);
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) part[(size_t)c * (N * S) + n * S + blockIdx.x] = make_float2(s1, s2);
}

// ms_bn_finalize + ms_bn_act(res_mode 0) in ONE launch, for a BatchNorm whose only consumer is its own activation (the encoder's code z_i and the code decoupler's z_s: two
// finalize -> apply pairs of ~5 us launches per step each doing microseconds of work).  Block (c, s): wave 0 Chan-merges channel c's statistics slots exactly as
// bn_finalize_kernel does (same order, fp64: the same record), block (c, 0) writes the record for the backward pass, and the block's 256 threads apply
// leaky(sc*u + sh) to images s, s + S, .. of the channel with bn_act_kernel's expression: the same bits as the two launches.
template <typename AT = float>
__global__ __launch_bounds__(kElemThreads) void bn_finalize_act_kernel(const float4* __restrict__ tab, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                       float4* __restrict__ coef, const void* __restrict__ u, void* __restrict__ out, int N, int C, int HW, int vec,
                                                                       float slope) {
  using IO = ActIO<AT>;
  __shared__ float2 sc_sh;
  const int c = blockIdx.x;
  if (threadIdx.x < 64) {
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
      const float sh = beta[c] - (float)mean * sc;
      if (blockIdx.y == 0) coef[c] = make_float4(sc, sh, (float)mean, invstd);
      sc_sh = make_float2(sc, sh);
    }
  }
  __syncthreads();
  const float2 cf = sc_sh;
  for (int n = blockIdx.y; n < N; n += gridDim.y) {
    const size_t pb = ((size_t)n * C + c) * HW;
    if (vec) {
      for (int i = threadIdx.x * 4; i < HW; i += kElemThreads * 4) {
        float4 t = IO::ld4(u, pb + i);
        t.x = leaky(cf.x * t.x + cf.y + 0.f, slope); t.y = leaky(cf.x * t.y + cf.y + 0.f, slope);
        t.z = leaky(cf.x * t.z + cf.y + 0.f, slope); t.w = leaky(cf.x * t.w + cf.y + 0.f, slope);
        IO::st4(out, pb + i, t);
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += kElemThreads) IO::st1(out, pb + i, leaky(cf.x * IO::ld1(u, pb + i) + cf.y + 0.f, slope));
    }
  }
}

// BatchNorm backward coefficients: du = sc*(g - mean(g) - uhat*mean(g*uhat)) = al*g + be*u + de
__global__ __launch_bounds__(64) void bn_bwd_coefs_kernel(const float2* __restrict__ part, int nparts, const float4* __restrict__ coef, double count,
                                                          float4* __restrict__ out) {
  // one wave per channel, shuffle reductions (a latency-floor kernel: see bn_finalize_kernel)
  const int c = blockIdx.x;
  // nparts == 0: `part` is the table of ms_conv2d_actbwd - [0] = {slots in use}, rows of kStatSlots from [1]
  const float2* row = (nparts == 0) ? part + 1 + (size_t)c * kStatSlots : part + (size_t)c * nparts;
  if (nparts == 0) nparts = (int)part[0].x;
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) { const float2 q = row[i]; s1 += (double)q.x; s2 += (double)q.y; }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (threadIdx.x == 0) {
    const float4 cf = coef[c];           // {sc, sh, mean, invstd}
    const double mean = cf.z, invstd = cf.w, sc = cf.x;
    const double c1 = s1 / count;
    const double c2 = s2 * invstd / count;
    const double be = -sc * c2 * invstd;
    out[c] = make_float4((float)sc, (float)be, (float)(-sc * c1 - be * mean), 0.f);
  }
}

// out[n,c,y,x] (+)= sum of the 2x2 block of in (gradient of nearest x2 up-sampling)
template <typename AT = float>
__global__ __launch_bounds__(kElemThreads) void pool2_sum_kernel(const void* __restrict__ in, void* __restrict__ out, int planes, int Ho, int Wo, int accumulate) {
  using IO = ActIO<AT>;
  const size_t total = (size_t)planes * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * kElemThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kElemThreads) {
    const int x = (int)(i % Wo);
    const size_t t = i / Wo;
    const int y = (int)(t % Ho);
    const size_t p = t / Ho;
    const size_t io = (p * 2 * Ho + 2 * y) * (size_t)(2 * Wo) + 2 * x;
    const float2 a = IO::ld2(in, io);
    const float2 b = IO::ld2(in, io + 2 * Wo);
    const float v = (a.x + a.y) + (b.x + b.y);
    IO::st1(out, i, accumulate ? IO::ld1(out, i) + v : v);
  }
}

// pool2_sum + accumulate + the output-activation backward of the block BELOW in one pass (ms_pool2_actbwd): the gradient w.r.t. the input of an up_type 'NN'
// block is pool2(d_hi) + (1x1 skip data-gradient); that tensor is the gradient w.r.t. the OUTPUT of the block below, whose backward starts with
// g = dout * lrelu'(act) and the two BatchNorm-backward sums (ms_act_bwd_reduce).  grid (S, N*C) as act_bwd_reduce; 4 low-resolution pixels per thread.
// POOL: a thread owns a 2x2 quad of OUTPUT pixels (a 4x4 patch of `in`) instead of 4 output pixels of a row, and also writes the quad's sum to `pooled`
// [N,C,Ho/2,Wo/2] = what ms_pool2_sum(gout) would write, in its order: the input of the NEXT block's 1x1 skip data-gradient, no pooling launch there.
template <typename AT = float, bool POOL = false>
__global__ __launch_bounds__(kElemThreads) void pool2_actbwd_kernel(const void* __restrict__ in, const void* __restrict__ add, void* __restrict__ gout,
                                                                    const void* __restrict__ act, const void* __restrict__ u, const float4* __restrict__ coef,
                                                                    float2* __restrict__ part, int C, int Ho, int Wo, int chunk, int S, int N, float slope,
                                                                    void* __restrict__ pooled, int in_lo) {
  // in_lo: `in` is ALREADY pooled ([N,C,Ho,Wo]: the conv in front stored 2x2 sums itself, MS_EPI_POOL2) - same arithmetic from there on
  __shared__ float red[16];
  const int p = blockIdx.y, c = p % C, n = p / C;
  const float mean = coef[c].z;
  const int HWo = Ho * Wo;
  const int beg = blockIdx.x * chunk, end = min(HWo, beg + chunk);
  const size_t base = (size_t)p * HWo;
  using IO = ActIO<AT>;
  const size_t ipb = (size_t)p * 4 * HWo;
  float s1 = 0.f, s2 = 0.f;
  for (int i = beg + threadIdx.x * 4; i < end; i += kElemThreads * 4) {
    if constexpr (POOL) {
      const int Wq = Wo >> 1, q = i >> 2;
      const int yq = q / Wq, xq = q - yq * Wq;                  // output pixels (2yq + r, 2xq + {0, 1}), r = 0, 1
      float2 v[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const size_t o = base + (size_t)(2 * yq + r) * Wo + 2 * xq;
        float2 t;
        if (in_lo) {
          t = IO::ld2(in, o);
        } else {
          const size_t r0 = ipb + (size_t)(4 * yq + 2 * r) * (2 * Wo) + 4 * xq;
          const float4 a0 = IO::ld4(in, r0), b0 = IO::ld4(in, r0 + 2 * Wo);
          t = make_float2((a0.x + a0.y) + (b0.x + b0.y), (a0.z + a0.w) + (b0.z + b0.w));
        }
        if (add != nullptr) { const float2 qq = IO::ld2(add, o); t.x = qq.x + t.x; t.y = qq.y + t.y; }
        const float2 rr = IO::ld2(act, o);
        const float2 uu = IO::ld2(u, o);
        t.x *= (rr.x > 0.f) ? 1.f : slope; t.y *= (rr.y > 0.f) ? 1.f : slope;
        IO::st2(gout, o, t);
        s1 += t.x + t.y;
        s2 += t.x * (uu.x - mean) + t.y * (uu.y - mean);
        v[r] = t;
      }
      auto rt = [](float x) { if constexpr (IO::kBytes == 2) return IO::up(ms_to_bf16(x)); else return x; };      // bf16 storage: the sum of the STORED values
      IO::st1(pooled, (size_t)p * (HWo >> 2) + q, (rt(v[0].x) + rt(v[0].y)) + (rt(v[1].x) + rt(v[1].y)));
    } else {
    const int y = i / Wo, x = i - y * Wo;                       // Wo % 4 == 0: the quad stays in one row
    float4 v;
    if (in_lo) {
      v = IO::ld4(in, base + i);
    } else {
      const size_t r0 = ipb + (size_t)(2 * y) * (2 * Wo) + 2 * x;
      const float4 a0 = IO::ld4(in, r0), a1 = IO::ld4(in, r0 + 4);
      const float4 b0 = IO::ld4(in, r0 + 2 * Wo), b1 = IO::ld4(in, r0 + 2 * Wo + 4);
      v = make_float4((a0.x + a0.y) + (b0.x + b0.y), (a0.z + a0.w) + (b0.z + b0.w), (a1.x + a1.y) + (b1.x + b1.y), (a1.z + a1.w) + (b1.z + b1.w));
    }
    if (add != nullptr) { const float4 q = IO::ld4(add, base + i); v.x = q.x + v.x; v.y = q.y + v.y; v.z = q.z + v.z; v.w = q.w + v.w; }
    const float4 r = IO::ld4(act, base + i);
    const float4 uu = IO::ld4(u, base + i);
    v.x *= (r.x > 0.f) ? 1.f : slope; v.y *= (r.y > 0.f) ? 1.f : slope; v.z *= (r.z > 0.f) ? 1.f : slope; v.w *= (r.w > 0.f) ? 1.f : slope;
    IO::st4(gout, base + i, v);
    s1 += (v.x + v.y) + (v.z + v.w);
    s2 += (v.x * (uu.x - mean) + v.y * (uu.y - mean)) + (v.z * (uu.z - mean) + v.w * (uu.w - mean));
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) part[(size_t)c * (N * S) + n * S + blockIdx.x] = make_float2(s1, s2);
}

// ---- heads (few output channels: VALU dot products, HBM-bound on h) -------------------------------------
constexpr int kMaxHeadC = 64, kMaxHeadK = 4;

// out[n,k,i] = sigmoid(b[k] + sum_c w[k][c]*h[n,c,i])
// st_mu != NULL (ms_head_fwd_styled): h is the INPUT of a MaxStyle layer that sits directly in front of the head; its output y = A/sig * (h - mu) + S (per plane,
// the expression and rounding of the layer's own kernels) is formed here and never written.
template <typename AT = float>
__global__ __launch_bounds__(kElemThreads) void head_sigmoid_kernel(const void* __restrict__ h, const float* __restrict__ w, const float* __restrict__ b,
                                                                    void* __restrict__ out, int C, int K, int HW, int apply_sigmoid,
                                                                    const float* __restrict__ st_mu, const float* __restrict__ st_sig, const float* __restrict__ st_A,
                                                                    const float* __restrict__ st_S) {
  using IO = ActIO<AT>;
  __shared__ float sw[kMaxHeadK * kMaxHeadC + kMaxHeadK];
  for (int i = threadIdx.x; i < K * C; i += kElemThreads) sw[i] = w[i];
  if (threadIdx.x < K) sw[kMaxHeadK * kMaxHeadC + threadIdx.x] = b ? b[threadIdx.x] : 0.f;
  const int n = blockIdx.y;
  __shared__ float sty[3 * kMaxHeadC];                 // {mu, A / sig, S} of the sample's planes
  const bool styled = st_mu != nullptr;
  if (styled && (int)threadIdx.x < C) {
    const int p = n * C + threadIdx.x;
    sty[threadIdx.x] = st_mu[p]; sty[kMaxHeadC + threadIdx.x] = st_A[p] / st_sig[p]; sty[2 * kMaxHeadC + threadIdx.x] = st_S[p];
  }
  __syncthreads();
  const size_t hb = (size_t)n * C * HW;
  for (int i = (blockIdx.x * kElemThreads + threadIdx.x) * 4; i < HW; i += gridDim.x * kElemThreads * 4) {
    float4 acc[kMaxHeadK];
#pragma unroll
    for (int k = 0; k < kMaxHeadK; ++k) { const float bb = sw[kMaxHeadK * kMaxHeadC + k]; acc[k] = make_float4(bb, bb, bb, bb); }
    for (int c = 0; c < C; ++c) {
      float4 v = IO::ld4(h, hb + (size_t)c * HW + i);
      if (styled) {
        const float m = sty[c], a = sty[kMaxHeadC + c], sh = sty[2 * kMaxHeadC + c];
        auto rt = [](float t) { if constexpr (IO::kBytes == 2) return IO::up(ms_to_bf16(t)); else return t; };      // bf16 storage: y as it would have been stored
        v.x = rt(a * (v.x - m) + sh); v.y = rt(a * (v.y - m) + sh); v.z = rt(a * (v.z - m) + sh); v.w = rt(a * (v.w - m) + sh);
      }
#pragma unroll
      for (int k = 0; k < kMaxHeadK; ++k) if (k < K) {
        const float ww = sw[k * C + c];
        acc[k].x += ww * v.x; acc[k].y += ww * v.y; acc[k].z += ww * v.z; acc[k].w += ww * v.w;
      }
    }
#pragma unroll
    for (int k = 0; k < kMaxHeadK; ++k) if (k < K) {
      float4 o = acc[k];
      if (apply_sigmoid) { o.x = 1.f / (1.f + expf(-o.x)); o.y = 1.f / (1.f + expf(-o.y)); o.z = 1.f / (1.f + expf(-o.z)); o.w = 1.f / (1.f + expf(-o.w)); }
      IO::st4(out, ((size_t)n * K + k) * HW + i, o);
    }
  }
}

// dh[n,c,i] = sum_k w[k][c] * dout[n,k,i] * out(1-out)
template <typename AT = float>
__global__ __launch_bounds__(kElemThreads) void head_sigmoid_bwd_kernel(const void* __restrict__ dout, const void* __restrict__ out, const float* __restrict__ w,
                                                                        void* __restrict__ dh, int C, int K, int HW, int apply_sigmoid) {
  using IO = ActIO<AT>;
  __shared__ float sw[kMaxHeadK * kMaxHeadC];
  for (int i = threadIdx.x; i < K * C; i += kElemThreads) sw[i] = w[i];
  __syncthreads();
  const int n = blockIdx.y;
  for (int i = (blockIdx.x * kElemThreads + threadIdx.x) * 4; i < HW; i += gridDim.x * kElemThreads * 4) {
    float4 d[kMaxHeadK];
#pragma unroll
    for (int k = 0; k < kMaxHeadK; ++k) if (k < K) {
      const float4 g = IO::ld4(dout, ((size_t)n * K + k) * HW + i);
      if (apply_sigmoid) {
        const float4 o = IO::ld4(out, ((size_t)n * K + k) * HW + i);
        d[k] = make_float4(g.x * (o.x * (1.f - o.x)), g.y * (o.y * (1.f - o.y)), g.z * (o.z * (1.f - o.z)), g.w * (o.w * (1.f - o.w)));
      } else d[k] = g;
    }
    for (int c = 0; c < C; ++c) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < kMaxHeadK; ++k) if (k < K) { const float ww = sw[k * C + c]; a.x += ww * d[k].x; a.y += ww * d[k].y; a.z += ww * d[k].z; a.w += ww * d[k].w; }
      IO::st4(dh, ((size_t)n * C + c) * HW + i, a);
    }
  }
}

// Segmentation head with fused loss and backward: logits = W h + b; logp = log_softmax; CE = -(1/M) sum logp[label];
// the loop maximises CE (loss = -CE, advanced_triplet...py:555), so d loss/d logit_k = -(softmax_k - 1[k==label]) / M.
// Writes dh = sum_k w[k][c]*dlogit_k, optional logits, and per-block partial sums of logp[label].
template <typename AT = float>
__global__ __launch_bounds__(kElemThreads) void head_ce_kernel(const void* __restrict__ h, const float* __restrict__ w, const float* __restrict__ b,
                                                               const int64_t* __restrict__ labels, void* __restrict__ dh, float* __restrict__ logits_out,
                                                               double* __restrict__ part, int C, int K, int HW, float grad_scale, const float* __restrict__ dscale) {
  using IO = ActIO<AT>;
  __shared__ float sw[kMaxHeadK * kMaxHeadC + kMaxHeadK];
  __shared__ double redd[16];
  if (dscale != nullptr) grad_scale *= *dscale;          // (ms_head_ce_ds: the upstream gradient as a device scalar - no host read of it)
  for (int i = threadIdx.x; i < K * C; i += kElemThreads) sw[i] = w[i];
  if (threadIdx.x < K) sw[kMaxHeadK * kMaxHeadC + threadIdx.x] = b ? b[threadIdx.x] : 0.f;
  __syncthreads();
  const int n = blockIdx.y;
  const size_t hb = (size_t)n * C * HW;
  double picked = 0.0;
  for (int i = blockIdx.x * kElemThreads + threadIdx.x; i < HW; i += gridDim.x * kElemThreads) {
    float z[kMaxHeadK];
#pragma unroll
    for (int k = 0; k < kMaxHeadK; ++k) z[k] = sw[kMaxHeadK * kMaxHeadC + k];
    for (int c = 0; c < C; ++c) {
      const float v = IO::ld1(h, hb + (size_t)c * HW + i);
#pragma unroll
      for (int k = 0; k < kMaxHeadK; ++k) if (k < K) z[k] = __builtin_fmaf(sw[k * C + c], v, z[k]);
    }
    float mx = z[0];
#pragma unroll
    for (int k = 1; k < kMaxHeadK; ++k) if (k < K) mx = fmaxf(mx, z[k]);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxHeadK; ++k) if (k < K) se += expf(z[k] - mx);
    const float lg = logf(se);
    const float lse = mx + lg;
    const int lab = (int)labels[(size_t)n * HW + i];
    {
      // picked log-softmax as (z_label - max) - log(sum): no cancellation against the rounded max + log(sum) (ulp(max) per pixel; config 4's first loss 7e-6 -> 9e-8 from
      // fp64).  Selected here so that max and log(sum) die before the class loop (two more live registers there halve head_ce_tail's occupancy: 46 -> 67 us)
      float zl = z[0];
#pragma unroll
      for (int k = 1; k < kMaxHeadK; ++k) if (k < K) zl = (k == lab) ? z[k] : zl;
      // a label outside [0, K) contributes nothing - neither to the loss nor (grad_scale below) to the gradient: F.nll_loss's ignore_index semantics (the reference
      // passes no such labels; until round 5 the class-0 term was picked for them.  ADVICE r5)
      picked += ((unsigned)lab < (unsigned)K) ? (double)((zl - mx) - lg) : 0.0;
    }
    float d[kMaxHeadK];
#pragma unroll
    for (int k = 0; k < kMaxHeadK; ++k) if (k < K) {
      const float pk = expf(z[k] - lse);
      d[k] = ((unsigned)lab < (unsigned)K ? grad_scale : 0.f) * (pk - (k == lab ? 1.f : 0.f));
      if (logits_out) logits_out[((size_t)n * K + k) * HW + i] = z[k];
    }
    if (dh) {
      for (int c = 0; c < C; ++c) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxHeadK; ++k) if (k < K) a = __builtin_fmaf(sw[k * C + c], d[k], a);
        IO::st1(dh, ((size_t)n * C + c) * HW + i, a);
      }
    }
  }
  picked = block_sum_d(picked, redd);
  if (threadIdx.x == 0) part[(size_t)n * gridDim.x + blockIdx.x] = picked;
}

// head_ce_kernel + the output-activation backward of the block whose output h is (ms_head_ce_actbwd; C <= 16): dh is multiplied by lrelu'(h) - sign(h) ==
// sign(pre-activation), and h is already in this kernel's hands - and the per-channel sums the BatchNorm backward of the block's last BatchNorm needs
// (sum g', sum g'*(u - mean_c)) are written per block in the partial layout of ms_act_bwd_reduce: bn_part[c][n*gridDim.x + blockIdx.x].
constexpr int kHeadFuseC = 16;
// VEC = 4: every thread owns 4 consecutive pixels (16-byte loads / stores of every channel plane; needs H*W % 4 == 0 and 16-byte aligned tensors)
template <int VEC, typename AT = float, bool PF = false>
__global__ __launch_bounds__(kElemThreads) void head_ce_actbwd_kernel(const void* __restrict__ h, const float* __restrict__ w, const float* __restrict__ b,
                                                                      const int64_t* __restrict__ labels, void* __restrict__ dh, double* __restrict__ part,
                                                                      int C, int K, int HW, float grad_scale,
                                                                      const void* __restrict__ bn_u, const float4* __restrict__ bn_coef, float2* __restrict__ bn_part, float slope) {
  using IO = ActIO<AT>;
  __shared__ float sw[kMaxHeadK * kHeadFuseC + kMaxHeadK];
  __shared__ float smean[kHeadFuseC];
  __shared__ double redd[16];
  for (int i = threadIdx.x; i < K * C; i += kElemThreads) sw[i] = w[i];
  if (threadIdx.x < K) sw[kMaxHeadK * kHeadFuseC + threadIdx.x] = b ? b[threadIdx.x] : 0.f;
  if (threadIdx.x < C) smean[threadIdx.x] = bn_coef[threadIdx.x].z;
  __syncthreads();
  const int n = blockIdx.y;
  const size_t hb = (size_t)n * C * HW;
  double picked = 0.0;
  float b1[kHeadFuseC], b2[kHeadFuseC];
#pragma unroll
  for (int c = 0; c < kHeadFuseC; ++c) { b1[c] = 0.f; b2[c] = 0.f; }
  for (int i = (blockIdx.x * kElemThreads + threadIdx.x) * VEC; i < HW; i += gridDim.x * kElemThreads * VEC) {
    float z[kMaxHeadK][VEC], hv[kHeadFuseC][VEC], uv[PF ? kHeadFuseC : 1][VEC];
    int labv[VEC];
    // every load of the item is issued before the first use (PF): h (C planes), the labels, and the raw BatchNorm input u that only the SECOND half of the
    // item needs.  Without it the grid - one item per thread, all waves in step - alternates between a load phase and an arithmetic phase twice per launch
    // (h ... softmax ... u ... stores): 201 MB in 68 us.  Costs 4*C more registers per thread.
#pragma unroll
    for (int c = 0; c < kHeadFuseC; ++c) {
      if (c < C) {
        if (VEC == 4) { const float4 t = IO::ld4(h, hb + (size_t)c * HW + i); hv[c][0] = t.x; hv[c][1 % VEC] = t.y; hv[c][2 % VEC] = t.z; hv[c][3 % VEC] = t.w; }
        else hv[c][0] = IO::ld1(h, hb + (size_t)c * HW + i);
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) labv[e] = (int)labels[(size_t)n * HW + i + e];
    if (PF) {
#pragma unroll
      for (int c = 0; c < kHeadFuseC; ++c) {
        if (c < C) {
          if (VEC == 4) { const float4 t = IO::ld4(bn_u, hb + (size_t)c * HW + i); uv[c][0] = t.x; uv[c][1 % VEC] = t.y; uv[c][2 % VEC] = t.z; uv[c][3 % VEC] = t.w; }
          else uv[c][0] = IO::ld1(bn_u, hb + (size_t)c * HW + i);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kMaxHeadK; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) z[k][e] = sw[kMaxHeadK * kHeadFuseC + k];
#pragma unroll
    for (int c = 0; c < kHeadFuseC; ++c) {
      if (c < C) {
#pragma unroll
        for (int k = 0; k < kMaxHeadK; ++k)
          if (k < K) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) z[k][e] = __builtin_fmaf(sw[k * C + c], hv[c][e], z[k][e]);
          }
      }
    }
    float d[kMaxHeadK][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float mx = z[0][e];
#pragma unroll
      for (int k = 1; k < kMaxHeadK; ++k) if (k < K) mx = fmaxf(mx, z[k][e]);
      float se = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxHeadK; ++k) if (k < K) se += expf(z[k][e] - mx);
      const float lg = logf(se);
      const float lse = mx + lg;
      const int lab = labv[e];
      {
        float zl = z[0][e];
#pragma unroll
        for (int k = 1; k < kMaxHeadK; ++k) if (k < K) zl = (k == lab) ? z[k][e] : zl;
        picked += ((unsigned)lab < (unsigned)K) ? (double)((zl - mx) - lg) : 0.0;      // (labels outside [0, K) are ignored: see head_ce_kernel)
      }
#pragma unroll
      for (int k = 0; k < kMaxHeadK; ++k) if (k < K) {
        const float pk = expf(z[k][e] - lse);
        d[k][e] = ((unsigned)lab < (unsigned)K ? grad_scale : 0.f) * (pk - (k == lab ? 1.f : 0.f));
      }
    }
#pragma unroll
    for (int c = 0; c < kHeadFuseC; ++c) {
      if (c < C) {
        float a[VEC], uu[VEC];
        if (PF) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) uu[e] = uv[c][e];
        } else if (VEC == 4) { const float4 t = IO::ld4(bn_u, hb + (size_t)c * HW + i); uu[0] = t.x; uu[1 % VEC] = t.y; uu[2 % VEC] = t.z; uu[3 % VEC] = t.w; }
        else uu[0] = IO::ld1(bn_u, hb + (size_t)c * HW + i);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float t = 0.f;
#pragma unroll
          for (int k = 0; k < kMaxHeadK; ++k) if (k < K) t = __builtin_fmaf(sw[k * C + c], d[k][e], t);
          t *= (hv[c][e] > 0.f) ? 1.f : slope;
          a[e] = t;
          b1[c] += t;
          b2[c] = __builtin_fmaf(t, uu[e] - smean[c], b2[c]);
        }
        if (VEC == 4) IO::st4(dh, ((size_t)n * C + c) * HW + i, make_float4(a[0], a[1 % VEC], a[2 % VEC], a[3 % VEC]));
        else IO::st1(dh, ((size_t)n * C + c) * HW + i, a[0]);
      }
    }
  }
  picked = block_sum_d(picked, redd);
  if (threadIdx.x == 0) part[(size_t)n * gridDim.x + blockIdx.x] = picked;
  // 2*C block sums with ONE barrier: wave-level shuffles first, then the four wave results through LDS (fixed order: deterministic)
  const int N = (int)gridDim.y, S = (int)gridDim.x;
  __shared__ float sred[kElemThreads / 64][2 * kHeadFuseC];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < kHeadFuseC; ++c) {
    if (c < C) {
      const float t1 = wave_sum(b1[c]), t2 = wave_sum(b2[c]);
      if (lane == 0) { sred[wv][c] = t1; sred[wv][kHeadFuseC + c] = t2; }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {
    const int c = threadIdx.x;
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int q = 0; q < kElemThreads / 64; ++q) { t1 += sred[q][c]; t2 += sred[q][kHeadFuseC + c]; }
    bn_part[(size_t)c * (N * S) + n * S + blockIdx.x] = make_float2(t1, t2);
  }
}

// ms_head_ce_tail: head_ce_actbwd_kernel<4> whose input h is not read but FORMED here - the output of the residual block in front of the head,
// h = lrelu((sc*u + sh) + skip[y/2][x/2]) from the block's second conv output u, that BatchNorm's record and the 1x1 skip conv at half resolution
// (ms_conv1x1_bnres's arithmetic, same order: the same bits).  The block output is never written; u is read ONCE for both halves of the item (h is consumed
// channel by channel into the logits, only its sign - 64 bits per thread - survives to the backward half, so the 64 registers of u fit beside the rest).
// POOL: a thread owns a 2x2 pixel quad (two 8-byte accesses per channel, ONE skip pixel) instead of 4 pixels of a row, and also writes the quad's sum of dh to
// `pooled` [N,C,H/2,W/2] = what ms_pool2_sum(dh) would write, (a.x + a.y) + (b.x + b.y): the next block's 1x1 skip data-gradient reads that, no pooling launch.
template <int K, bool POOL, typename AT = float>
__global__ __launch_bounds__(kElemThreads) void head_ce_tail_kernel(const void* __restrict__ u, const void* __restrict__ skip, const float4* __restrict__ bn_coef,
                                                                    const float* __restrict__ w, const float* __restrict__ b, const int64_t* __restrict__ labels,
                                                                    void* __restrict__ dh, double* __restrict__ part, int HW, int W, float grad_scale,
                                                                    float2* __restrict__ bn_part, float slope, void* __restrict__ pooled) {
  using IO = ActIO<AT>;
  constexpr int C = kHeadFuseC;             // compile-time channel count: no per-channel branches, the K*C weights arrive as wide scalar loads
  // head weights, bias and the BatchNorm record are read through UNIFORM addresses (scalar loads): no LDS copy and no vector registers holding K*C weights
  // across the item; every tensor access is a uniform per-(sample, channel) base + a 32-bit per-thread byte offset (global_load saddr form: no 64-bit
  // per-channel address pairs in vector registers)
  __shared__ double redd[16];
  const int n = blockIdx.y;
  const int HWl = HW >> 2, Wl = W >> 1;
  const char* ub = reinterpret_cast<const char*>(u) + (size_t)n * C * HW * IO::kBytes;
  const char* kb = reinterpret_cast<const char*>(skip) + (size_t)n * C * HWl * IO::kBytes;
  char* db = reinterpret_cast<char*>(dh) + (size_t)n * C * HW * IO::kBytes;
  const size_t pu = (size_t)HW * IO::kBytes, pk = (size_t)HWl * IO::kBytes;       // plane strides in bytes
  double picked = 0.0;
  float b1[C], b2[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { b1[c] = 0.f; b2[c] = 0.f; }
  for (int i = (blockIdx.x * kElemThreads + threadIdx.x) * 4; i < HW; i += gridDim.x * kElemThreads * 4) {
    // !POOL: i = first of 4 pixels of one row (W % 4 == 0; skip pixels x/2, x/2 + 1 of row y/2).  POOL: i / 4 = the low-resolution pixel (yl, xl) whose 2x2 quad this is;
    // element e of tu / labv / a below = pixel (2yl + (e >> 1), 2xl + (e & 1))
    int y, x; unsigned ou, ou2 = 0u, ok;
    if constexpr (POOL) { const int q = i >> 2; y = q / Wl; x = q - y * Wl; ou = (unsigned)(2 * y * W + 2 * x) * IO::kBytes; ou2 = ou + (unsigned)W * IO::kBytes; ok = (unsigned)q * IO::kBytes; }
    else { y = i / W; x = i - y * W; ou = (unsigned)i * IO::kBytes; ok = (unsigned)((y >> 1) * Wl + (x >> 1)) * IO::kBytes; }
    float4 tu[C]; float2 ts[C];
    int labv[4];
    if constexpr (POOL) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float2 r0 = IO::ld2(ub + c * pu + ou, 0), r1 = IO::ld2(ub + c * pu + ou2, 0);
        const float sv = IO::ld1(kb + c * pk + ok, 0);
        tu[c] = make_float4(r0.x, r0.y, r1.x, r1.y); ts[c] = make_float2(sv, sv);
      }
      const int64_t* lp = labels + (size_t)n * HW + (size_t)(2 * y) * W + 2 * x;
      labv[0] = (int)lp[0]; labv[1] = (int)lp[1]; labv[2] = (int)lp[W]; labv[3] = (int)lp[W + 1];
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) { tu[c] = IO::ld4(ub + c * pu + ou, 0); ts[c] = IO::ld2(kb + c * pk + ok, 0); }
#pragma unroll
      for (int e = 0; e < 4; ++e) labv[e] = (int)labels[(size_t)n * HW + i + e];
    }
    float z[K][4];
    unsigned pos[2] = {0u, 0u};                                 // bit 4*(c & 7) + e of word c >> 3: h[c][e] > 0
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) z[k][e] = b != nullptr ? b[k] : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float4 cf = bn_coef[c];
      const float hv[4] = {leaky((cf.x * tu[c].x + cf.y) + ts[c].x, slope), leaky((cf.x * tu[c].y + cf.y) + ts[c].x, slope),
                           leaky((cf.x * tu[c].z + cf.y) + ts[c].y, slope), leaky((cf.x * tu[c].w + cf.y) + ts[c].y, slope)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pos[c >> 3] |= (hv[e] > 0.f ? 1u : 0u) << (4 * (c & 7) + e);
#pragma unroll
        for (int k = 0; k < K; ++k) z[k][e] = __builtin_fmaf(w[k * C + c], hv[e], z[k][e]);
      }
    }
    float d[K][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float mx = z[0][e];
#pragma unroll
      for (int k = 1; k < K; ++k) mx = fmaxf(mx, z[k][e]);
      float se = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) se += expf(z[k][e] - mx);
      const float lg = logf(se);
      const float lse = mx + lg;
      const int lab = labv[e];
      {
        float zl = z[0][e];
#pragma unroll
        for (int k = 1; k < K; ++k) zl = (k == lab) ? z[k][e] : zl;
        picked += ((unsigned)lab < (unsigned)K) ? (double)((zl - mx) - lg) : 0.0;      // (labels outside [0, K) are ignored: see head_ce_kernel)
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float pk_ = expf(z[k][e] - lse);
        d[k][e] = ((unsigned)lab < (unsigned)K ? grad_scale : 0.f) * (pk_ - (k == lab ? 1.f : 0.f));
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float mean = bn_coef[c].z;
      const float uq[4] = {tu[c].x, tu[c].y, tu[c].z, tu[c].w};
      float a[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) t = __builtin_fmaf(w[k * C + c], d[k][e], t);
        t *= ((pos[c >> 3] >> (4 * (c & 7) + e)) & 1u) ? 1.f : slope;
        a[e] = t;
        b1[c] += t;
        b2[c] = __builtin_fmaf(t, uq[e] - mean, b2[c]);
      }
      if constexpr (POOL) {
        IO::st2(db + c * pu + ou, 0, make_float2(a[0], a[1]));
        IO::st2(db + c * pu + ou2, 0, make_float2(a[2], a[3]));
        // (bf16 storage: the sum of the STORED values, like ms_pool2_sum reading dh back)
        auto rt = [](float v) { if constexpr (IO::kBytes == 2) return IO::up(ms_to_bf16(v)); else return v; };
        IO::st1(reinterpret_cast<char*>(pooled) + ((size_t)n * C + c) * pk + ok, 0, (rt(a[0]) + rt(a[1])) + (rt(a[2]) + rt(a[3])));
      } else {
        IO::st4(db + c * pu + ou, 0, make_float4(a[0], a[1], a[2], a[3]));
      }
    }
  }
  picked = block_sum_d(picked, redd);
  if (threadIdx.x == 0) part[(size_t)n * gridDim.x + blockIdx.x] = picked;
  const int N = (int)gridDim.y, S = (int)gridDim.x;
  __shared__ float sred[kElemThreads / 64][2 * C];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float t1 = wave_sum(b1[c]), t2 = wave_sum(b2[c]);
    if (lane == 0) { sred[wv][c] = t1; sred[wv][C + c] = t2; }
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {
    const int c = threadIdx.x;
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int q = 0; q < kElemThreads / 64; ++q) { t1 += sred[q][c]; t2 += sred[q][C + c]; }
    bn_part[(size_t)c * (N * S) + n * S + blockIdx.x] = make_float2(t1, t2);
  }
}

__global__ __launch_bounds__(256) void ce_finalize_kernel(const double* __restrict__ part, int nparts, double scale, float* __restrict__ loss_out,
                                                          const int* __restrict__ slot_dev) {
  __shared__ double redd[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  s = block_sum_d(s, redd);
  if (threadIdx.x == 0) loss_out[slot_dev ? *slot_dev : 0] = (float)(s * scale);
}

// Confusion matrix of argmax(logits) vs labels for Dice / IoU (common_utils/metrics.py:12-52,134-245 use a numpy confusion matrix /
// medpy.metric.binary.dc = 2|A n B| / (|A|+|B|)): per-block LDS histogram, one integer atomic per non-zero bin per block.
__global__ __launch_bounds__(kElemThreads) void confusion_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                                unsigned long long* __restrict__ cm, int K, int HW) {
  __shared__ unsigned int hist[kMaxHeadK * kMaxHeadK];
  if (threadIdx.x < kMaxHeadK * kMaxHeadK) hist[threadIdx.x] = 0;
  __syncthreads();
  const int n = blockIdx.y;
  const float* lp = logits + (size_t)n * K * HW;
  for (int i = blockIdx.x * kElemThreads + threadIdx.x; i < HW; i += gridDim.x * kElemThreads) {
    int best = 0; float bv = lp[i];
    for (int k = 1; k < K; ++k) { const float v = lp[(size_t)k * HW + i]; if (v > bv) { bv = v; best = k; } }     // first maximum, as torch.argmax
    const int lab = (int)labels[(size_t)n * HW + i];
    if (lab >= 0 && lab < K) atomicAdd(&hist[lab * K + best], 1u);
  }
  __syncthreads();
  if (threadIdx.x < K * K && hist[threadIdx.x] != 0) atomicAdd(&cm[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

// per-plane min-max rescale (common_utils/basic_operations.py:257-281): y = (x - min) / (max - min + eps) * (hi - lo) + lo
__global__ __launch_bounds__(1024) void rescale_intensity_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, float lo, float hi, float eps) {
  __shared__ float smin[16], smax[16];
  const float* xp = x + (size_t)blockIdx.x * HW;
  float* yp = y + (size_t)blockIdx.x * HW;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < HW; i += 1024) { const float v = xp[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off, 64)); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
  __syncthreads();
  mn = smin[0]; mx = smax[0];
  for (int w = 1; w < 16; ++w) { mn = fminf(mn, smin[w]); mx = fmaxf(mx, smax[w]); }
  const float den = mx - mn + eps, span = hi - lo;
  for (int i = threadIdx.x; i < HW; i += 1024) yp[i] = (xp[i] - mn) / den * span + lo;
}

}  // namespace ms

using namespace ms;

extern "C" int ms_rescale_intensity(const float* x, float* y, int planes, int HW, float new_min, float new_max, float eps, void* stream) {
  if (planes < 1 || HW < 1) { set_error("ms_rescale_intensity: invalid shape"); return MS_ERR_INVALID; }
  MS_LAUNCH(rescale_intensity_kernel, dim3(planes), dim3(1024), 0, (hipStream_t)stream, x, y, HW, new_min, new_max, eps);
  return check_launch("rescale_intensity");
}

extern "C" int ms_confusion(const float* logits, const int64_t* labels, unsigned long long* cm, int N, int K, int HW, void* stream) {
  if (N < 1 || K < 1 || K > kMaxHeadK || HW < 1 || N > 65535) { set_error("ms_confusion: unsupported shape (K <= %d)", kMaxHeadK); return MS_ERR_INVALID; }
  dim3 grid(std::min(cdiv(HW, kElemThreads), 64), N);
  MS_LAUNCH(confusion_kernel, grid, dim3(kElemThreads), 0, (hipStream_t)stream, logits, labels, cm, K, HW);
  return check_launch("confusion");
}

template <typename AT>
static int bn_act_impl(const void* u, const float* coef4, const void* res, int res_mode, void* out, int N, int C, int H, int W, float slope, void* stream) {
  if (N < 1 || C < 1 || H < 1 || W < 1 || res_mode < 0 || res_mode > 2) { set_error("ms_bn_act: invalid argument"); return MS_ERR_INVALID; }
  if (!(slope >= 0.f && slope <= 1.f)) { set_error("ms_bn_act: activation slope outside [0, 1]"); return MS_ERR_INVALID; }
  if (res_mode != 0 && res == nullptr) { set_error("ms_bn_act: residual missing"); return MS_ERR_INVALID; }
  if (res_mode == 2 && ((H | W) & 1)) { set_error("ms_bn_act: half-resolution residual needs even H,W"); return MS_ERR_INVALID; }
  if ((long)N * C > 65535) { set_error("ms_bn_act: too many planes"); return MS_ERR_INVALID; }
  const ElemSplit sp = elem_split(N * C, H * W);
  dim3 grid(sp.S, N * C), block(kElemThreads);
  hipStream_t st = (hipStream_t)stream;
  const float4* cf = (const float4*)coef4;
  if (res_mode == 0) MS_LAUNCH((bn_act_kernel<0, AT>), grid, block, 0, st, u, cf, res, out, C, H, W, sp.chunk, slope);
  else if (res_mode == 1) MS_LAUNCH((bn_act_kernel<1, AT>), grid, block, 0, st, u, cf, res, out, C, H, W, sp.chunk, slope);
  else MS_LAUNCH((bn_act_kernel<2, AT>), grid, block, 0, st, u, cf, res, out, C, H, W, sp.chunk, slope);
  return check_launch("bn_act");
}
extern "C" int ms_bn_act(const float* u, const float* coef4, const float* res, int res_mode, float* out, int N, int C, int H, int W, float slope, void* stream) {
  return bn_act_impl<float>(u, coef4, res, res_mode, out, N, C, H, W, slope, stream);
}
// `_bf16` twins of the streaming kernels: activation tensors (u, res, out, gradients, h, ...) are bf16 bit patterns, everything else as in the fp32 entry point
extern "C" int ms_bn_act_bf16(const uint16_t* u, const float* coef4, const uint16_t* res, int res_mode, uint16_t* out, int N, int C, int H, int W, float slope, void* stream) {
  return bn_act_impl<ms_bf16>(u, coef4, res, res_mode, out, N, C, H, W, slope, stream);
}

template <typename AT>
static int bn_finalize_act_impl(const float* stats, int nparts, const float* gamma, const float* beta, float eps, float* coef4, const void* u, void* out,
                                int N, int C, int H, int W, float slope, void* stream) {
  if (N < 1 || C < 1 || H < 1 || W < 1 || nparts != kStatSlots || stats == nullptr || gamma == nullptr || beta == nullptr || coef4 == nullptr || u == nullptr || out == nullptr) {
    set_error("ms_bn_finalize_act: invalid argument (nparts must be ms_conv_stats_parts())"); return MS_ERR_INVALID;
  }
  if (!(slope >= 0.f && slope <= 1.f)) { set_error("ms_bn_finalize_act: slope in [0, 1]"); return MS_ERR_INVALID; }
  if (!aligned16(stats) || !aligned16(coef4)) { set_error("ms_bn_finalize_act: stats / coef4 must be 16-byte aligned"); return MS_ERR_ALIGN; }
  const int HW = H * W;
  const int vec = (W % 4 == 0 && aligned16(u) && aligned16(out)) ? 1 : 0;          // bn_act_kernel's rule (same expression either way)
  const int S = std::max(1, std::min(N, cdiv(2 * num_cus(), C)));                   // enough blocks for the chip at 128 channels; every block re-reads its channel's slots (<= 8 KB, L2)
  MS_LAUNCH((bn_finalize_act_kernel<AT>), dim3(C, S), dim3(kElemThreads), 0, (hipStream_t)stream, (const float4*)stats, gamma, beta, eps, (float4*)coef4, u, out, N, C, HW, vec, slope);
  return check_launch("bn_finalize_act");
}
extern "C" int ms_bn_finalize_act(const float* stats, int nparts, const float* gamma, const float* beta, float eps, float* coef4, const float* u, float* out,
                                  int N, int C, int H, int W, float slope, void* stream) {
  return bn_finalize_act_impl<float>(stats, nparts, gamma, beta, eps, coef4, u, out, N, C, H, W, slope, stream);
}
extern "C" int ms_bn_finalize_act_bf16(const float* stats, int nparts, const float* gamma, const float* beta, float eps, float* coef4, const uint16_t* u, uint16_t* out,
                                       int N, int C, int H, int W, float slope, void* stream) {
  return bn_finalize_act_impl<ms_bf16>(stats, nparts, gamma, beta, eps, coef4, u, out, N, C, H, W, slope, stream);
}

extern "C" int ms_act_bwd_parts(int N, int C, int HW) { return N * elem_split(N * C, HW).S; }

template <typename AT>
static int act_bwd_reduce_impl(const void* gin, const void* ref, const void* u, const float* coef4, void* gout, float* part2,
                               int N, int C, int HW, float slope, void* stream) {
  if (N < 1 || C < 1 || HW < 1) { set_error("ms_act_bwd_reduce: invalid shape"); return MS_ERR_INVALID; }
  if ((long)N * C > 65535) { set_error("ms_act_bwd_reduce: too many planes"); return MS_ERR_INVALID; }
  const ElemSplit sp = elem_split(N * C, HW);
  dim3 grid(sp.S, N * C), block(kElemThreads);
  hipStream_t st = (hipStream_t)stream;
  if (ref != nullptr) MS_LAUNCH((act_bwd_reduce_kernel<0, AT>), grid, block, 0, st, gin, ref, u, (const float4*)coef4, gout, (float2*)part2, C, HW, sp.chunk, sp.S, N, slope);
  else MS_LAUNCH((act_bwd_reduce_kernel<1, AT>), grid, block, 0, st, gin, ref, u, (const float4*)coef4, gout, (float2*)part2, C, HW, sp.chunk, sp.S, N, slope);
  return check_launch("act_bwd_reduce");
}
extern "C" int ms_act_bwd_reduce(const float* gin, const float* ref, const float* u, const float* coef4, float* gout, float* part2,
                                 int N, int C, int HW, float slope, void* stream) {
  return act_bwd_reduce_impl<float>(gin, ref, u, coef4, gout, part2, N, C, HW, slope, stream);
}
extern "C" int ms_act_bwd_reduce_bf16(const uint16_t* gin, const uint16_t* ref, const uint16_t* u, const float* coef4, uint16_t* gout, float* part2,
                                      int N, int C, int HW, float slope, void* stream) {
  return act_bwd_reduce_impl<ms_bf16>(gin, ref, u, coef4, gout, part2, N, C, HW, slope, stream);
}

extern "C" int ms_bn_bwd_coefs(const float* part2, int nparts, const float* coef4, double count, float* coef_out4, int C, void* stream) {
  if (C < 1 || nparts < 0 || count <= 0) { set_error("ms_bn_bwd_coefs: invalid argument"); return MS_ERR_INVALID; }
  MS_LAUNCH(bn_bwd_coefs_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, (const float2*)part2, nparts, (const float4*)coef4, count, (float4*)coef_out4);
  return check_launch("bn_bwd_coefs");
}

template <typename AT>
static int pool2_sum_impl(const void* in, void* out, int planes, int Ho, int Wo, int accumulate, void* stream) {
  if (planes < 1 || Ho < 1 || Wo < 1) { set_error("ms_pool2_sum: invalid shape"); return MS_ERR_INVALID; }
  const size_t total = (size_t)planes * Ho * Wo;
  const int blocks = (int)std::min<size_t>((total + kElemThreads - 1) / kElemThreads, 4096);
  MS_LAUNCH((pool2_sum_kernel<AT>), dim3(blocks), dim3(kElemThreads), 0, (hipStream_t)stream, in, out, planes, Ho, Wo, accumulate);
  return check_launch("pool2_sum");
}
extern "C" int ms_pool2_sum(const float* in, float* out, int planes, int Ho, int Wo, int accumulate, void* stream) { return pool2_sum_impl<float>(in, out, planes, Ho, Wo, accumulate, stream); }
extern "C" int ms_pool2_sum_bf16(const uint16_t* in, uint16_t* out, int planes, int Ho, int Wo, int accumulate, void* stream) { return pool2_sum_impl<ms_bf16>(in, out, planes, Ho, Wo, accumulate, stream); }

// out = (pool2_sum(in) [+ add]) * lrelu'(act), part2 = the partial sums of ms_act_bwd_reduce ([C][ms_act_bwd_parts(N,C,Ho*Wo)][2]).  in [N,C,2Ho,2Wo]; add (may be NULL
// or == out), out, act, u [N,C,Ho,Wo]; Wo % 4 == 0, 16-byte aligned.
template <typename AT>
static int pool2_actbwd_impl(const void* in, const void* add, void* out, const void* act, const void* u, const float* coef4, float* part2,
                             int N, int C, int Ho, int Wo, float slope, void* stream, void* pooled = nullptr, int in_lo = 0) {
  if (N < 1 || C < 1 || Ho < 1 || Wo < 4 || Wo % 4 != 0) { set_error("ms_pool2_actbwd: invalid shape (Wo %% 4 == 0)"); return MS_ERR_INVALID; }
  if ((long)N * C > 65535) { set_error("ms_pool2_actbwd: too many planes"); return MS_ERR_INVALID; }
  if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(u) |
        reinterpret_cast<uintptr_t>(coef4) | reinterpret_cast<uintptr_t>(add)) & 15u) != 0) { set_error("ms_pool2_actbwd: tensors must be 16-byte aligned"); return MS_ERR_ALIGN; }
  const ElemSplit sp = elem_split(N * C, Ho * Wo);
  if (pooled != nullptr) {
    if (Ho % 2 != 0 || (reinterpret_cast<uintptr_t>(pooled) & 3u) != 0) { set_error("ms_pool2_actbwd_pool: even Ho, 4-byte aligned pooled"); return MS_ERR_INVALID; }
    MS_LAUNCH((pool2_actbwd_kernel<AT, true>), dim3(sp.S, N * C), dim3(kElemThreads), 0, (hipStream_t)stream, in, add, out, act, u, (const float4*)coef4, (float2*)part2,
              C, Ho, Wo, sp.chunk, sp.S, N, slope, pooled, in_lo);
    return check_launch("pool2_actbwd_pool");
  }
  MS_LAUNCH((pool2_actbwd_kernel<AT, false>), dim3(sp.S, N * C), dim3(kElemThreads), 0, (hipStream_t)stream, in, add, out, act, u, (const float4*)coef4, (float2*)part2,
            C, Ho, Wo, sp.chunk, sp.S, N, slope, (void*)nullptr, in_lo);
  return check_launch("pool2_actbwd");
}
extern "C" int ms_pool2_actbwd(const float* in, const float* add, float* out, const float* act, const float* u, const float* coef4, float* part2,
                               int N, int C, int Ho, int Wo, float slope, void* stream) {
  return pool2_actbwd_impl<float>(in, add, out, act, u, coef4, part2, N, C, Ho, Wo, slope, stream);
}
// ms_pool2_actbwd that also writes pooled [N,C,Ho/2,Wo/2] = ms_pool2_sum(out) (same bits): see the kernel.  Ho even, Wo % 4 == 0.
extern "C" int ms_pool2_actbwd_pool(const float* in, const float* add, float* out, const float* act, const float* u, const float* coef4, float* part2,
                                    int N, int C, int Ho, int Wo, float slope, float* pooled, void* stream) {
  if (pooled == nullptr) { set_error("ms_pool2_actbwd_pool: pooled is required"); return MS_ERR_INVALID; }
  return pool2_actbwd_impl<float>(in, add, out, act, u, coef4, part2, N, C, Ho, Wo, slope, stream, pooled);
}
extern "C" int ms_pool2_actbwd_pool_bf16(const uint16_t* in, const uint16_t* add, uint16_t* out, const uint16_t* act, const uint16_t* u, const float* coef4, float* part2,
                                         int N, int C, int Ho, int Wo, float slope, uint16_t* pooled, void* stream) {
  if (pooled == nullptr) { set_error("ms_pool2_actbwd_pool: pooled is required"); return MS_ERR_INVALID; }
  return pool2_actbwd_impl<ms_bf16>(in, add, out, act, u, coef4, part2, N, C, Ho, Wo, slope, stream, pooled);
}
// ms_pool2_actbwd[_pool] whose `in_lo` [N,C,Ho,Wo] is already the pooled tensor (the conv in front stored the 2x2 sums itself: MS_EPI_POOL2); pooled may be NULL.
extern "C" int ms_add_actbwd(const float* in_lo, const float* add, float* out, const float* act, const float* u, const float* coef4, float* part2,
                             int N, int C, int Ho, int Wo, float slope, float* pooled, void* stream) {
  return pool2_actbwd_impl<float>(in_lo, add, out, act, u, coef4, part2, N, C, Ho, Wo, slope, stream, pooled, 1);
}
extern "C" int ms_add_actbwd_bf16(const uint16_t* in_lo, const uint16_t* add, uint16_t* out, const uint16_t* act, const uint16_t* u, const float* coef4, float* part2,
                                  int N, int C, int Ho, int Wo, float slope, uint16_t* pooled, void* stream) {
  return pool2_actbwd_impl<ms_bf16>(in_lo, add, out, act, u, coef4, part2, N, C, Ho, Wo, slope, stream, pooled, 1);
}
extern "C" int ms_pool2_actbwd_bf16(const uint16_t* in, const uint16_t* add, uint16_t* out, const uint16_t* act, const uint16_t* u, const float* coef4, float* part2,
                                    int N, int C, int Ho, int Wo, float slope, void* stream) {
  return pool2_actbwd_impl<ms_bf16>(in, add, out, act, u, coef4, part2, N, C, Ho, Wo, slope, stream);
}

static int head_check(int N, int C, int K, int HW, const char* who) {
  if (N < 1 || C < 1 || C > kMaxHeadC || K < 1 || K > kMaxHeadK || HW < 1) { set_error("%s: unsupported head shape C=%d (<=%d) K=%d (<=%d)", who, C, kMaxHeadC, K, kMaxHeadK); return MS_ERR_INVALID; }
  if (N > 65535) { set_error("%s: batch too large", who); return MS_ERR_INVALID; }
  return MS_OK;
}

template <typename AT>
static int head_fwd_impl(const void* h, const float* w, const float* b, void* out, int N, int C, int K, int HW, int apply_sigmoid, void* stream,
                         const float* st_mu = nullptr, const float* st_sig = nullptr, const float* st_A = nullptr, const float* st_S = nullptr) {
  if (int e = head_check(N, C, K, HW, "ms_head_fwd")) return e;
  if (HW % 4 != 0) { set_error("ms_head_fwd: H*W must be a multiple of 4"); return MS_ERR_INVALID; }
  dim3 grid(std::min(cdiv(HW, kElemThreads * 4), 256), N);
  MS_LAUNCH((head_sigmoid_kernel<AT>), grid, dim3(kElemThreads), 0, (hipStream_t)stream, h, w, b, out, C, K, HW, apply_sigmoid, st_mu, st_sig, st_A, st_S);
  return check_launch("head_fwd");
}
// ms_style_fwd's restyle + ms_head_fwd in one pass: x [N,C,HW] is the INPUT of a MaxStyle layer directly in front of the 1x1 head, (mu, sig, coefA, coefS) [N*C] what
// ms_style_fwd (called with y = NULL) left for it; y = coefA/sig * (x - mu) + coefS is formed per element (same expression, same rounding) and never written.
extern "C" int ms_head_fwd_styled(const float* x, const float* mu, const float* sig, const float* coefA, const float* coefS, const float* w, const float* b, float* out,
                                  int N, int C, int K, int HW, int apply_sigmoid, void* stream) {
  if (mu == nullptr || sig == nullptr || coefA == nullptr || coefS == nullptr) { set_error("ms_head_fwd_styled: the layer's statistics are required"); return MS_ERR_INVALID; }
  return head_fwd_impl<float>(x, w, b, out, N, C, K, HW, apply_sigmoid, stream, mu, sig, coefA, coefS);
}
extern "C" int ms_head_fwd_styled_bf16(const uint16_t* x, const float* mu, const float* sig, const float* coefA, const float* coefS, const float* w, const float* b, uint16_t* out,
                                       int N, int C, int K, int HW, int apply_sigmoid, void* stream) {
  if (mu == nullptr || sig == nullptr || coefA == nullptr || coefS == nullptr) { set_error("ms_head_fwd_styled: the layer's statistics are required"); return MS_ERR_INVALID; }
  return head_fwd_impl<ms_bf16>(x, w, b, out, N, C, K, HW, apply_sigmoid, stream, mu, sig, coefA, coefS);
}
extern "C" int ms_head_fwd(const float* h, const float* w, const float* b, float* out, int N, int C, int K, int HW, int apply_sigmoid, void* stream) {
  return head_fwd_impl<float>(h, w, b, out, N, C, K, HW, apply_sigmoid, stream);
}
extern "C" int ms_head_fwd_bf16(const uint16_t* h, const float* w, const float* b, uint16_t* out, int N, int C, int K, int HW, int apply_sigmoid, void* stream) {
  return head_fwd_impl<ms_bf16>(h, w, b, out, N, C, K, HW, apply_sigmoid, stream);
}

template <typename AT>
static int head_bwd_impl(const void* dout, const void* out, const float* w, void* dh, int N, int C, int K, int HW, int apply_sigmoid, void* stream) {
  if (int e = head_check(N, C, K, HW, "ms_head_bwd")) return e;
  if (HW % 4 != 0) { set_error("ms_head_bwd: H*W must be a multiple of 4"); return MS_ERR_INVALID; }
  dim3 grid(std::min(cdiv(HW, kElemThreads * 4), 256), N);
  MS_LAUNCH((head_sigmoid_bwd_kernel<AT>), grid, dim3(kElemThreads), 0, (hipStream_t)stream, dout, out, w, dh, C, K, HW, apply_sigmoid);
  return check_launch("head_bwd");
}
extern "C" int ms_head_bwd(const float* dout, const float* out, const float* w, float* dh, int N, int C, int K, int HW, int apply_sigmoid, void* stream) {
  return head_bwd_impl<float>(dout, out, w, dh, N, C, K, HW, apply_sigmoid, stream);
}
extern "C" int ms_head_bwd_bf16(const uint16_t* dout, const uint16_t* out, const float* w, uint16_t* dh, int N, int C, int K, int HW, int apply_sigmoid, void* stream) {
  return head_bwd_impl<ms_bf16>(dout, out, w, dh, N, C, K, HW, apply_sigmoid, stream);
}

extern "C" size_t ms_head_ce_ws_bytes(int N, int HW) { return (size_t)N * std::min(cdiv(HW, kElemThreads), 256) * sizeof(double) + 64; }

template <typename AT>
static int head_ce_impl(const void* h, const float* w, const float* b, const int64_t* labels, void* dh, float* logits, float* loss_out,
                        const int* loss_slot_dev, int N, int C, int K, int HW, float loss_sign, void* ws, size_t ws_bytes, void* stream, const float* dscale = nullptr) {
  if (int e = head_check(N, C, K, HW, "ms_head_ce")) return e;
  if (ws == nullptr || ws_bytes < ms_head_ce_ws_bytes(N, HW)) { set_error("ms_head_ce: workspace too small"); return MS_ERR_WORKSPACE; }
  const int gx = std::min(cdiv(HW, kElemThreads), 256);
  dim3 grid(gx, N);
  const double M = (double)N * HW;
  // loss = loss_sign * CE, CE = -(1/M) sum logp[label];  d loss / d logit_k = loss_sign * (p_k - 1[k==label]) / M
  MS_LAUNCH((head_ce_kernel<AT>), grid, dim3(kElemThreads), 0, (hipStream_t)stream, h, w, b, labels, dh, logits, (double*)ws, C, K, HW, (float)(loss_sign / M), dscale);
  if (int e = check_launch("head_ce")) return e;
  MS_LAUNCH(ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, gx * N, -(double)loss_sign / M, loss_out, loss_slot_dev);
  return check_launch("ce_finalize");
}
extern "C" int ms_head_ce(const float* h, const float* w, const float* b, const int64_t* labels, float* dh, float* logits, float* loss_out,
                          const int* loss_slot_dev, int N, int C, int K, int HW, float loss_sign, void* ws, size_t ws_bytes, void* stream) {
  return head_ce_impl<float>(h, w, b, labels, dh, logits, loss_out, loss_slot_dev, N, C, K, HW, loss_sign, ws, ws_bytes, stream);
}
extern "C" int ms_head_ce_ds(const float* h, const float* w, const float* b, const int64_t* labels, float* dh, float* logits, float* loss_out,
                             const int* loss_slot_dev, int N, int C, int K, int HW, float loss_sign, const float* grad_scale_dev, void* ws, size_t ws_bytes, void* stream) {
  return head_ce_impl<float>(h, w, b, labels, dh, logits, loss_out, loss_slot_dev, N, C, K, HW, loss_sign, ws, ws_bytes, stream, grad_scale_dev);
}
// (logits stay fp32: they are an output for the caller, not an activation of the loop)
extern "C" int ms_head_ce_bf16(const uint16_t* h, const float* w, const float* b, const int64_t* labels, uint16_t* dh, float* logits, float* loss_out,
                               const int* loss_slot_dev, int N, int C, int K, int HW, float loss_sign, void* ws, size_t ws_bytes, void* stream) {
  return head_ce_impl<ms_bf16>(h, w, b, labels, dh, logits, loss_out, loss_slot_dev, N, C, K, HW, loss_sign, ws, ws_bytes, stream);
}

static int head_fuse_gx(int HW) { return std::max(1, std::min(cdiv(HW, kElemThreads * 4), 256)); }      // 4 pixels per thread: 16-byte accesses, the per-block reduction amortised
extern "C" int ms_head_ce_actbwd_parts(int N, int C, int HW) { return (C >= 1 && C <= kHeadFuseC) ? N * head_fuse_gx(HW) : 0; }

// ms_head_ce (loss + dh) whose dh is already masked by the activation that produced h, with the BatchNorm-backward sums of that block
// (h = lrelu(bn(bn_u) + skip): encoder_decoder.py:344-346): replaces ms_head_ce + ms_act_bwd_reduce.  C <= 16 (ms_head_ce_actbwd_parts returns 0 otherwise).
template <typename AT>
static int head_ce_actbwd_impl(const void* h, const float* w, const float* b, const int64_t* labels, void* dh, float* loss_out, const int* loss_slot_dev,
                               int N, int C, int K, int HW, float loss_sign, void* ws, size_t ws_bytes,
                               const void* bn_u, const float* bn_coef4, float* bn_part, float act_slope, void* stream) {
  if (int e = head_check(N, C, K, HW, "ms_head_ce_actbwd")) return e;
  if (C > kHeadFuseC || dh == nullptr || bn_u == nullptr || bn_coef4 == nullptr || bn_part == nullptr || !aligned16(bn_coef4)) {
    set_error("ms_head_ce_actbwd: C <= %d, dh, bn_u, bn_coef4 (16-byte aligned) and bn_part are required", kHeadFuseC); return MS_ERR_INVALID;
  }
  if (ws == nullptr || ws_bytes < ms_head_ce_ws_bytes(N, HW)) { set_error("ms_head_ce_actbwd: workspace too small"); return MS_ERR_WORKSPACE; }
  const int gx = head_fuse_gx(HW);
  const double M = (double)N * HW;
  const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(dh) | reinterpret_cast<uintptr_t>(bn_u)) & 15u) == 0;
  // (a variant with every load of an item issued up front - 239 instead of 175 VGPRs - measured null, 69.0 vs 69.3 us at 16x16x256x256, profiles/r03_experiments.txt: removed in round 5)
  if (vec) MS_LAUNCH((head_ce_actbwd_kernel<4, AT>), dim3(gx, N), dim3(kElemThreads), 0, (hipStream_t)stream, h, w, b, labels, dh, (double*)ws, C, K, HW, (float)(loss_sign / M),
                     bn_u, (const float4*)bn_coef4, (float2*)bn_part, act_slope);
  else MS_LAUNCH((head_ce_actbwd_kernel<1, AT>), dim3(gx, N), dim3(kElemThreads), 0, (hipStream_t)stream, h, w, b, labels, dh, (double*)ws, C, K, HW, (float)(loss_sign / M),
                 bn_u, (const float4*)bn_coef4, (float2*)bn_part, act_slope);
  if (int e = check_launch("head_ce_actbwd")) return e;
  if (loss_out == nullptr) return MS_OK;               // the caller sums the gx*N partials in `ws` itself (ms_step_tail)
  MS_LAUNCH(ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, gx * N, -(double)loss_sign / M, loss_out, loss_slot_dev);
  return check_launch("ce_finalize");
}
extern "C" int ms_head_ce_actbwd(const float* h, const float* w, const float* b, const int64_t* labels, float* dh, float* loss_out, const int* loss_slot_dev,
                                 int N, int C, int K, int HW, float loss_sign, void* ws, size_t ws_bytes,
                                 const float* bn_u, const float* bn_coef4, float* bn_part, float act_slope, void* stream) {
  return head_ce_actbwd_impl<float>(h, w, b, labels, dh, loss_out, loss_slot_dev, N, C, K, HW, loss_sign, ws, ws_bytes, bn_u, bn_coef4, bn_part, act_slope, stream);
}
// ms_conv1x1_bnres (mode 5) + ms_head_ce_actbwd in one pass over the block's second conv output: the output of the LAST residual block of the segmentation
// decoder (encoder_decoder.py:344-346, up_type 'NN') is formed inside the head kernel as lrelu(bn(u) + skip[y/2][x/2]) - u [N,C,H,W] that block's second conv
// output, coef4 its BatchNorm record, skip [N,C,H/2,W/2] the 1x1 skip conv (+ bias) at half resolution - and is never written.  Everything else as
// ms_head_ce_actbwd (dh already masked by lrelu', bn_part the BatchNorm-backward sums, loss_out may be NULL for ms_step_tail).  C == 16 (FCN_16's last block), K <= 4.
template <typename AT>
static int head_ce_tail_impl(const void* u, const void* skip, const float* coef4, const float* w, const float* b, const int64_t* labels, void* dh, float* loss_out,
                             const int* loss_slot_dev, int N, int C, int K, int H, int W, float loss_sign, void* ws, size_t ws_bytes, float* bn_part, float act_slope, void* pooled,
                             void* stream) {
  const int HW = H * W;
  if (int e = head_check(N, C, K, HW, "ms_head_ce_tail")) return e;
  if (C != kHeadFuseC || K > kMaxHeadK || u == nullptr || skip == nullptr || coef4 == nullptr || dh == nullptr || bn_part == nullptr || W < 4 || W % 4 != 0 || H % 2 != 0 || !aligned16(u) || !aligned16(dh) ||
      !aligned16(coef4) || (reinterpret_cast<uintptr_t>(skip) & 7u) != 0 || (reinterpret_cast<uintptr_t>(pooled) & 3u) != 0 || !(act_slope >= 0.f && act_slope <= 1.f)) {
    set_error("ms_head_ce_tail: C == %d, W %% 4 == 0, even H, 16-byte aligned u / dh / coef4, 8-byte aligned skip, slope in [0, 1]", kHeadFuseC); return MS_ERR_INVALID;
  }
  if (ws == nullptr || ws_bytes < ms_head_ce_ws_bytes(N, HW)) { set_error("ms_head_ce_tail: workspace too small"); return MS_ERR_WORKSPACE; }
  const int gx = head_fuse_gx(HW);
  const double M = (double)N * HW;
#define MS_HT(KK, PP) MS_LAUNCH((head_ce_tail_kernel<KK, PP, AT>), dim3(gx, N), dim3(kElemThreads), 0, (hipStream_t)stream, u, skip, (const float4*)coef4, w, b, labels, dh, (double*)ws, HW, W, \
                               (float)(loss_sign / M), (float2*)bn_part, act_slope, pooled)
  if (pooled != nullptr) { switch (K) { case 1: MS_HT(1, true); break; case 2: MS_HT(2, true); break; case 3: MS_HT(3, true); break; default: MS_HT(4, true); } }
  else { switch (K) { case 1: MS_HT(1, false); break; case 2: MS_HT(2, false); break; case 3: MS_HT(3, false); break; default: MS_HT(4, false); } }
#undef MS_HT
  if (int e = check_launch("head_ce_tail")) return e;
  if (loss_out == nullptr) return MS_OK;
  MS_LAUNCH(ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, gx * N, -(double)loss_sign / M, loss_out, loss_slot_dev);
  return check_launch("ce_finalize");
}
extern "C" int ms_head_ce_tail(const float* u, const float* skip, const float* coef4, const float* w, const float* b, const int64_t* labels, float* dh, float* loss_out,
                               const int* loss_slot_dev, int N, int C, int K, int H, int W, float loss_sign, void* ws, size_t ws_bytes, float* bn_part, float act_slope, float* pooled,
                               void* stream) {
  return head_ce_tail_impl<float>(u, skip, coef4, w, b, labels, dh, loss_out, loss_slot_dev, N, C, K, H, W, loss_sign, ws, ws_bytes, bn_part, act_slope, pooled, stream);
}
extern "C" int ms_head_ce_tail_bf16(const uint16_t* u, const uint16_t* skip, const float* coef4, const float* w, const float* b, const int64_t* labels, uint16_t* dh, float* loss_out,
                                    const int* loss_slot_dev, int N, int C, int K, int H, int W, float loss_sign, void* ws, size_t ws_bytes, float* bn_part, float act_slope, uint16_t* pooled,
                                    void* stream) {
  return head_ce_tail_impl<ms_bf16>(u, skip, coef4, w, b, labels, dh, loss_out, loss_slot_dev, N, C, K, H, W, loss_sign, ws, ws_bytes, bn_part, act_slope, pooled, stream);
}
extern "C" int ms_head_ce_actbwd_bf16(const uint16_t* h, const float* w, const float* b, const int64_t* labels, uint16_t* dh, float* loss_out, const int* loss_slot_dev,
                                      int N, int C, int K, int HW, float loss_sign, void* ws, size_t ws_bytes,
                                      const uint16_t* bn_u, const float* bn_coef4, float* bn_part, float act_slope, void* stream) {
  return head_ce_actbwd_impl<ms_bf16>(h, w, b, labels, dh, loss_out, loss_slot_dev, N, C, K, HW, loss_sign, ws, ws_bytes, bn_u, bn_coef4, bn_part, act_slope, stream);
}
