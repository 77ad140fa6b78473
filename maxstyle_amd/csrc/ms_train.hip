// Small kernels of the OUTER update: what loss.backward() + AdamW.step() add to the inner-loop kernels
// (train_adv_supervised_segmentation_triplet.py:532-535; advanced_triplet_recon_segmentation_model.py:718-729, 731-786, 1055-1086).
//   ms_bn_bwd_full      BatchNorm backward coefficients + weight.grad / bias.grad of the BatchNorm + bias.grad of the skip conv
//   ms_channel_sum      bias.grad of a convolution whose output gradient is materialised
//   ms_head_wgrad       weight.grad / bias.grad of the 1x1 heads (segmentation logits + cross entropy, image + sigmoid + MSE)
//   ms_mse_loss         0.5 * MSELoss(reduction='mean') and its gradient
//   ms_adamw_step       torch.optim.AdamW / Adam on one flat buffer
//   ms_bn_running_update  running_mean / running_var of a tracking BatchNorm forward
#include <algorithm>
#include "ms_common.h"
#include "maxstyle_hip.h"

namespace ms {

// du = al*g + be*u + de (see ms_bn_bwd_coefs) and, from the same two sums S1 = sum g, S2 = sum g*u:
//   bias.grad = S1,  weight.grad = sum g*uhat = S2*invstd        (S2 = sum g*(u - mean): centred by ms_act_bwd_reduce)
__global__ __launch_bounds__(256) void bn_bwd_full_kernel(const float2* __restrict__ part, int nparts, const float4* __restrict__ coef, double count,
                                                          float4* __restrict__ out, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ dsum, int accumulate) {
  __shared__ double redd[16];
  const int c = blockIdx.x;
  // nparts == 0: `part` is the table of ms_conv2d_actbwd - [0] = {slots in use}, rows of kStatSlots from [1]
  const float2* row = (nparts == 0) ? part + 1 + (size_t)c * kStatSlots : part + (size_t)c * nparts;
  if (nparts == 0) nparts = (int)part[0].x;
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) { const float2 q = row[i]; s1 += (double)q.x; s2 += (double)q.y; }
  s1 = block_sum_d(s1, redd);
  s2 = block_sum_d(s2, redd);
  if (threadIdx.x == 0) {
    const float4 cf = coef[c];           // {sc, sh, mean, invstd}
    const double mean = cf.z, invstd = cf.w, sc = cf.x;
    const double c1 = s1 / count;
    const double c2 = s2 * invstd / count;                     // s2 is the centred sum (see act_bwd_reduce_kernel)
    const double be = -sc * c2 * invstd;
    if (out) out[c] = make_float4((float)sc, (float)be, (float)(-sc * c1 - be * mean), 0.f);
    const float dg = (float)(s2 * invstd), db = (float)s1;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + dg : dg;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + db : db;
    if (dsum) dsum[c] = accumulate ? dsum[c] + db : db;
  }
}

// out[c] (+)= sum over n, hw of x[n,c,hw].  Stage 1: one workgroup per (channel, image) plane -> part[c][n] (fp64); stage 2: fixed-order sum.
__global__ __launch_bounds__(256) void channel_sum_part_kernel(const float* __restrict__ x, int C, int HW, double* __restrict__ part) {
  __shared__ double redd[16];
  const int c = blockIdx.x, n = blockIdx.y;
  const float* p = x + ((size_t)n * C + c) * HW;
  float s0 = 0.f, s1 = 0.f;
  if (HW % 4 == 0) {
    int i = threadIdx.x * 4;
    for (; i + 1024 < HW; i += 2048) {
      const float4 v = *reinterpret_cast<const float4*>(p + i), w = *reinterpret_cast<const float4*>(p + i + 1024);
      s0 += (v.x + v.y) + (v.z + v.w); s1 += (w.x + w.y) + (w.z + w.w);
    }
    for (; i < HW; i += 1024) { const float4 v = *reinterpret_cast<const float4*>(p + i); s0 += (v.x + v.y) + (v.z + v.w); }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) s0 += p[i];
  }
  const double acc = block_sum_d((double)s0 + (double)s1, redd);
  if (threadIdx.x == 0) part[(size_t)c * gridDim.y + n] = acc;
}

__global__ __launch_bounds__(64) void channel_sum_final_kernel(const double* __restrict__ part, int N, float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 64) s += part[(size_t)c * N + i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)s : (float)s;
}

constexpr int kHeadMaxC = 64, kHeadMaxK = 4, kHeadCB = 16;

// partial[block][k*(C+1) + c] = sum_i d[k][i]*h[c][i]  (c == C: sum_i d[k][i] -> bias.grad)
//   MODE 0: d = scale*(softmax(aux)_k - 1[k == label])   aux = logits [N,K,HW]
//   MODE 1: d = scale*(aux - target)*aux*(1-aux)          aux = sigmoid output [N,K,HW]
//   MODE 2: d = scale*aux
template <int MODE>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ h, const float* __restrict__ aux, const void* __restrict__ tgt, float scale,
                                                         float* __restrict__ partial, int C, int K, int HW, int chunk, const float* __restrict__ dscale) {
  __shared__ float red[4][kHeadMaxK * (kHeadCB + 1)];
  if (dscale != nullptr) scale *= *dscale;               // (ms_head_wgrad_ds: the upstream gradient as a device scalar)
  const int n = blockIdx.y;
  const int beg = blockIdx.x * chunk, end = min(HW, beg + chunk);
  const float* hp = h + (size_t)n * C * HW;
  const float* ap = aux + (size_t)n * K * HW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* outp = partial + ((size_t)n * gridDim.x + blockIdx.x) * (size_t)(K * (C + 1));
  for (int c0 = 0; c0 < C; c0 += kHeadCB) {
    const bool last = (c0 + kHeadCB >= C);
    float acc[kHeadMaxK][kHeadCB + 1];
#pragma unroll
    for (int k = 0; k < kHeadMaxK; ++k)
#pragma unroll
      for (int c = 0; c <= kHeadCB; ++c) acc[k][c] = 0.f;
    for (int i = beg + threadIdx.x; i < end; i += 256) {
      float d[kHeadMaxK];
      if (MODE == 0) {
        float z[kHeadMaxK], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < kHeadMaxK; ++k) if (k < K) { z[k] = ap[(size_t)k * HW + i]; mx = fmaxf(mx, z[k]); }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < kHeadMaxK; ++k) if (k < K) se += expf(z[k] - mx);
        const float lse = mx + logf(se);
        const int lab = (int)reinterpret_cast<const int64_t*>(tgt)[(size_t)n * HW + i];
#pragma unroll
        for (int k = 0; k < kHeadMaxK; ++k) d[k] = (k < K) ? scale * (expf(z[k] - lse) - (k == lab ? 1.f : 0.f)) : 0.f;
      } else if (MODE == 1) {
        const float* tp = reinterpret_cast<const float*>(tgt) + (size_t)n * K * HW;
#pragma unroll
        for (int k = 0; k < kHeadMaxK; ++k) {
          d[k] = 0.f;
          if (k < K) { const float o = ap[(size_t)k * HW + i]; d[k] = scale * (o - tp[(size_t)k * HW + i]) * (o * (1.f - o)); }
        }
      } else {
#pragma unroll
        for (int k = 0; k < kHeadMaxK; ++k) d[k] = (k < K) ? scale * ap[(size_t)k * HW + i] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < kHeadCB; ++c) {
        const float v = (c0 + c < C) ? hp[(size_t)(c0 + c) * HW + i] : 0.f;
#pragma unroll
        for (int k = 0; k < kHeadMaxK; ++k) acc[k][c] += d[k] * v;
      }
#pragma unroll
      for (int k = 0; k < kHeadMaxK; ++k) acc[k][kHeadCB] += d[k];
    }
#pragma unroll
    for (int k = 0; k < kHeadMaxK; ++k)
#pragma unroll
      for (int c = 0; c <= kHeadCB; ++c) {
        const float s = wave_sum(acc[k][c]);
        if (lane == 0) red[wave][k * (kHeadCB + 1) + c] = s;
      }
    __syncthreads();
    if (threadIdx.x < kHeadMaxK * (kHeadCB + 1)) {
      const int k = threadIdx.x / (kHeadCB + 1), c = threadIdx.x % (kHeadCB + 1);
      const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
      if (k < K) {
        if (c < kHeadCB) { if (c0 + c < C) outp[k * (C + 1) + c0 + c] = s; }
        else if (last) outp[k * (C + 1) + C] = s;
      }
    }
    __syncthreads();
  }
}

// dw[k][c] (+)= sum_slots partial[slot][k*(C+1)+c];  db[k] (+)= ...[k*(C+1)+C]      (fixed order, fp64)
__global__ __launch_bounds__(64) void head_wgrad_reduce_kernel(const float* __restrict__ partial, int nslots, int C, int K, float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  const int idx = blockIdx.x;            // k*(C+1)+c
  const int k = idx / (C + 1), c = idx % (C + 1);
  double s = 0.0;
  for (int i = threadIdx.x; i < nslots; i += 64) s += (double)partial[(size_t)i * (K * (C + 1)) + idx];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    float* dst = (c < C) ? dw + k * C + c : (db ? db + k : nullptr);
    if (dst) *dst = accumulate ? *dst + (float)s : (float)s;
  }
}

// partial sums of (x-t)^2 in fp64; dx = grad_scale*(x-t)
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ x, const float* __restrict__ t, size_t n, float grad_scale, float* __restrict__ dx, double* __restrict__ part,
                                                  const float* __restrict__ dscale) {
  __shared__ double redd[16];
  if (dscale != nullptr) grad_scale *= *dscale;          // (ms_mse_loss_ds)
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = x[i] - t[i];
    s += (double)d * (double)d;
    if (dx) dx[i] = grad_scale * d;
  }
  s = block_sum_d(s, redd);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sum_finalize_kernel(const double* __restrict__ part, int nparts, double scale, float* __restrict__ out) {
  __shared__ double redd[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  s = block_sum_d(s, redd);
  if (threadIdx.x == 0) *out = (float)(s * scale);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                                    float lr, float b1, float b2, float eps, float wd, int step, const int* __restrict__ step_dev) {
  const int t = step_dev ? (*step_dev + 1) : step;
  const double bc1 = 1.0 - pow((double)b1, (double)t);
  const double bc2 = 1.0 - pow((double)b2, (double)t);
  const float step_size = (float)((double)lr / bc1);
  const float rs = (float)sqrt(bc2);
  const float decay = 1.f - lr * wd;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = m[i] * b1 + gi * (1.f - b1);
    const float vi = v[i] * b2 + gi * gi * (1.f - b2);
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / rs + eps;
    p[i] = p[i] * decay - step_size * (mi / denom);
  }
}

// nn.BatchNorm2d training forward with track_running_stats: running = (1-mom)*running + mom*batch (unbiased variance)
__global__ void bn_running_kernel(const float4* __restrict__ coef, float* __restrict__ rm, float* __restrict__ rv, int C, float count, float mom, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = coef[c].z, invstd = coef[c].w;
  float var = 1.f / (invstd * invstd) - eps;
  if (count > 1.f) var *= count / (count - 1.f);
  rm[c] = rm[c] * (1.f - mom) + mom * mean;
  rv[c] = rv[c] * (1.f - mom) + mom * var;
}

// One launch re-packs every convolution weight of the three sub-nets from the flat parameter buffer into the two kernel layouts
// (ms_conv2d forward / data-gradient).  Source-driven: thread = one weight element; the zero padding of the packed buffers is never touched.
struct RepackDesc {          // 64 bytes, device-resident table built once by the host (addresses are stable)
  long long begin;           // prefix sum: first global element index of this tensor
  long long src_off;         // float offset into the flat buffer
  float* dst_f; float* dst_d;
  int kind;                  // 0: Conv2d [d0=Cout][d1=Cin][k][k], 1: ConvTranspose2d(k=2,s=2) [d0=Cin][d1=Cout][2][2]
  int d0, d1, k;
  int cinp_f, coutp_f, cinp_d, coutp_d;
};

__global__ __launch_bounds__(256) void repack_kernel(const float* __restrict__ flat, const RepackDesc* __restrict__ desc, int ndesc, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].begin <= i) lo = mid; else hi = mid - 1; }
  const RepackDesc d = desc[lo];
  const int e = (int)(i - d.begin);
  const float v = flat[d.src_off + e];
  const int kk = d.k * d.k;
  const int tap = e % kk, ky = tap / d.k, kx = tap % d.k;
  const int b = (e / kk) % d.d1, a = e / (kk * d.d1);
  if (d.kind == 0) {         // a = co, b = ci
    d.dst_f[((size_t)tap * d.cinp_f + b) * d.coutp_f + a] = v;
    const int tapr = (d.k - 1 - ky) * d.k + (d.k - 1 - kx);
    d.dst_d[((size_t)tapr * d.cinp_d + a) * d.coutp_d + b] = v;
  } else {                   // a = ci, b = co, tap = dy*2+dx
    d.dst_f[(size_t)a * d.coutp_f + tap * d.d1 + b] = v;
    d.dst_d[((size_t)tap * d.cinp_d + b) * d.coutp_d + a] = v;
  }
}

// the same for every BatchNorm of a pass in one launch: blockIdx.x = layer
struct BnRunDesc { const float4* coef; float* rm; float* rv; int C; float count; };
__global__ __launch_bounds__(256) void bn_running_batch_kernel(const BnRunDesc* __restrict__ desc, float mom, float eps) {
  const BnRunDesc d = desc[blockIdx.x];
  for (int c = threadIdx.x; c < d.C; c += 256) {
    const float mean = d.coef[c].z, invstd = d.coef[c].w;
    float var = 1.f / (invstd * invstd) - eps;
    if (d.count > 1.f) var *= d.count / (d.count - 1.f);
    d.rm[c] = d.rm[c] * (1.f - mom) + mom * mean;
    d.rv[c] = d.rv[c] * (1.f - mom) + mom * var;
  }
}

}  // namespace ms
using namespace ms;

extern "C" size_t ms_bn_running_desc_bytes(void) { return sizeof(BnRunDesc); }

extern "C" int ms_bn_running_update_batch(const void* desc_dev, int nlayers, float momentum, float eps, void* stream) {
  if (nlayers < 1 || desc_dev == nullptr) { set_error("ms_bn_running_update_batch: nothing to do"); return MS_ERR_INVALID; }
  MS_LAUNCH(bn_running_batch_kernel, dim3(nlayers), dim3(256), 0, (hipStream_t)stream, (const BnRunDesc*)desc_dev, momentum, eps);
  return check_launch("bn_running_update_batch");
}

extern "C" size_t ms_repack_desc_bytes(void) { return sizeof(RepackDesc); }

extern "C" int ms_repack_weights(const float* flat, const void* desc_dev, int ndesc, long long total, void* stream) {
  if (ndesc < 1 || total < 1) { set_error("ms_repack_weights: nothing to do"); return MS_ERR_INVALID; }
  MS_LAUNCH(repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat, (const RepackDesc*)desc_dev, ndesc, total);
  return check_launch("repack");
}

extern "C" int ms_bn_bwd_full(const float* part2, int nparts, const float* coef4, double count, float* coef_out4, float* dgamma, float* dbeta, float* dsum,
                              int accumulate, int C, void* stream) {
  if (C < 1 || nparts < 0 || count <= 0) { set_error("ms_bn_bwd_full: invalid argument"); return MS_ERR_INVALID; }
  MS_LAUNCH(bn_bwd_full_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, (const float2*)part2, nparts, (const float4*)coef4, count, (float4*)coef_out4,
            dgamma, dbeta, dsum, accumulate);
  return check_launch("bn_bwd_full");
}

extern "C" size_t ms_channel_sum_ws_bytes(int N, int C) { return (size_t)N * C * sizeof(double); }

extern "C" int ms_channel_sum(const float* x, int N, int C, int HW, float* out, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (N < 1 || C < 1 || HW < 1 || N > 65535) { set_error("ms_channel_sum: invalid shape"); return MS_ERR_INVALID; }
  if (ws == nullptr || ws_bytes < ms_channel_sum_ws_bytes(N, C)) { set_error("ms_channel_sum: workspace too small"); return MS_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  MS_LAUNCH(channel_sum_part_kernel, dim3(C, N), dim3(256), 0, st, x, C, HW, (double*)ws);
  if (int e = check_launch("channel_sum_part")) return e;
  MS_LAUNCH(channel_sum_final_kernel, dim3(C), dim3(64), 0, st, (const double*)ws, N, out, accumulate);
  return check_launch("channel_sum_final");
}

static int head_wgrad_chunk(int HW) { return std::max(1024, cdiv(cdiv(HW, 64), 256) * 256); }     // <= 64 blocks per image
extern "C" size_t ms_head_wgrad_ws_bytes(int N, int C, int K, int HW) {
  return (size_t)N * cdiv(HW, head_wgrad_chunk(HW)) * K * (C + 1) * sizeof(float);
}

static int head_wgrad_impl(const float* h, const float* aux, const void* target, int mode, float scale, float* dw, float* db,
                           int N, int C, int K, int HW, int accumulate, void* ws, size_t ws_bytes, void* stream, const float* dscale) {
  if (N < 1 || C < 1 || C > kHeadMaxC || K < 1 || K > kHeadMaxK || HW < 1 || N > 65535) { set_error("ms_head_wgrad: unsupported head shape C=%d K=%d", C, K); return MS_ERR_INVALID; }
  if (mode < 0 || mode > 2 || (mode != 2 && target == nullptr)) { set_error("ms_head_wgrad: invalid mode / missing target"); return MS_ERR_INVALID; }
  if (ws == nullptr || ws_bytes < ms_head_wgrad_ws_bytes(N, C, K, HW)) { set_error("ms_head_wgrad: workspace too small"); return MS_ERR_WORKSPACE; }
  const int chunk = head_wgrad_chunk(HW);
  dim3 grid(cdiv(HW, chunk), N);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) MS_LAUNCH(head_wgrad_kernel<0>, grid, dim3(256), 0, st, h, aux, target, scale, (float*)ws, C, K, HW, chunk, dscale);
  else if (mode == 1) MS_LAUNCH(head_wgrad_kernel<1>, grid, dim3(256), 0, st, h, aux, target, scale, (float*)ws, C, K, HW, chunk, dscale);
  else MS_LAUNCH(head_wgrad_kernel<2>, grid, dim3(256), 0, st, h, aux, target, scale, (float*)ws, C, K, HW, chunk, dscale);
  if (int e = check_launch("head_wgrad")) return e;
  MS_LAUNCH(head_wgrad_reduce_kernel, dim3(K * (C + 1)), dim3(64), 0, st, (const float*)ws, (int)(grid.x * N), C, K, dw, db, accumulate);
  return check_launch("head_wgrad_reduce");
}

extern "C" int ms_head_wgrad(const float* h, const float* aux, const void* target, int mode, float scale, float* dw, float* db,
                             int N, int C, int K, int HW, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  return head_wgrad_impl(h, aux, target, mode, scale, dw, db, N, C, K, HW, accumulate, ws, ws_bytes, stream, nullptr);
}
extern "C" int ms_head_wgrad_ds(const float* h, const float* aux, const void* target, int mode, float scale, const float* scale_dev, float* dw, float* db,
                                int N, int C, int K, int HW, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  return head_wgrad_impl(h, aux, target, mode, scale, dw, db, N, C, K, HW, accumulate, ws, ws_bytes, stream, scale_dev);
}

extern "C" size_t ms_mse_ws_bytes(void) { return 1024 * sizeof(double); }

static int mse_loss_impl(const float* x, const float* target, size_t n, float loss_scale, float grad_scale, float* loss_out, float* dx,
                         void* ws, size_t ws_bytes, void* stream, const float* dscale) {
  if (n < 1) { set_error("ms_mse_loss: empty input"); return MS_ERR_INVALID; }
  if (ws == nullptr || ws_bytes < ms_mse_ws_bytes()) { set_error("ms_mse_loss: workspace too small"); return MS_ERR_WORKSPACE; }
  const int nb = (int)std::min<size_t>(1024, (n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  MS_LAUNCH(mse_kernel, dim3(nb), dim3(256), 0, st, x, target, n, grad_scale, dx, (double*)ws, dscale);
  if (int e = check_launch("mse")) return e;
  if (loss_out) {
    MS_LAUNCH(sum_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nb, (double)loss_scale, loss_out);
    return check_launch("mse_finalize");
  }
  return MS_OK;
}
extern "C" int ms_mse_loss(const float* x, const float* target, size_t n, float loss_scale, float grad_scale, float* loss_out, float* dx,
                           void* ws, size_t ws_bytes, void* stream) {
  return mse_loss_impl(x, target, n, loss_scale, grad_scale, loss_out, dx, ws, ws_bytes, stream, nullptr);
}
extern "C" int ms_mse_loss_ds(const float* x, const float* target, size_t n, float loss_scale, float grad_scale, const float* grad_scale_dev, float* loss_out, float* dx,
                              void* ws, size_t ws_bytes, void* stream) {
  return mse_loss_impl(x, target, n, loss_scale, grad_scale, loss_out, dx, ws, ws_bytes, stream, grad_scale_dev);
}

extern "C" int ms_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float weight_decay,
                             int step, const int* step_dev, void* stream) {
  if (n < 1 || (step < 1 && step_dev == nullptr)) { set_error("ms_adamw_step: invalid argument"); return MS_ERR_INVALID; }
  const int nb = (int)std::min<size_t>(4096, (n + 255) / 256);
  MS_LAUNCH(adamw_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2, eps, weight_decay, step, step_dev);
  return check_launch("adamw");
}

extern "C" int ms_bn_running_update(const float* coef4, float* running_mean, float* running_var, int C, double count, float momentum, float eps, void* stream) {
  if (C < 1 || count < 1) { set_error("ms_bn_running_update: invalid argument"); return MS_ERR_INVALID; }
  MS_LAUNCH(bn_running_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, (const float4*)coef4, running_mean, running_var, C, (float)count, momentum, eps);
  return check_launch("bn_running_update");
}
