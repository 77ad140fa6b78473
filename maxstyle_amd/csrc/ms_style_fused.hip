// Single-read fused MaxStyle forward for gfx950: moments + style mixing + restyle with x read from HBM exactly once.
//
// Reference semantics: /root/reference/src/advanced/maxstyle.py:157-188 (same arithmetic as ms_style.hip).
//
// The restyle of plane (b,c) needs the statistics of plane (perm[b],c) and - on the first call - the batch standard
// deviation over all B planes of channel c, so a naive fusion needs a grid-wide barrier.  Here the dependency is kept
// per CHANNEL: work units (channel c, sample b, chunk s) are numbered channel-major and workgroup w processes units
// w, w+grid, ... in increasing order; it keeps its chunk of x in REGISTERS (up to 64 floats per thread), publishes the
// chunk's (mean, M2) as two tagged 8-byte granules {tag, value} with agent-scope (write-through) stores, then polls the
// granules of its channel's B*S units until every tag is set - the data IS the flag, one memory round trip -, merges the
// partials (Chan, fp64), computes its plane's affine coefficients and writes y from registers.
// HBM traffic = 4 B/element read + 4 B/element written = the algorithmic 8 B/element.
//
// Progress: the grid never exceeds the workgroups that fit the chip together (1 x 1024 threads or 4 x 256 threads per CU at
// <= 96 VGPRs), and a workgroup only ever waits for units of its own channel, which are the current or an earlier unit of
// other workgroups of the same launch (channel-major numbering + increasing processing order): the lowest unfinished channel
// always has every unit either done or being loaded, so it completes and releases its waiters.  Every spin is bounded (error
// word) - a mis-sized launch flags an error instead of hanging the GPU.  Visibility: 8-byte agent-scope atomics on both sides
// (MI355X_MICROARCH.md "Valid forms", R2: the granule carries its own tag).
// Tags are LAUNCH EPOCHS: the state block holds an epoch word; every workgroup reads it at start (T = epoch + 1), publishes and accepts
// only granules tagged T, and the last workgroup to finish stores epoch = T.  Nothing is re-initialised between launches - an earlier
// version cleared the table with hipMemsetAsync before every launch, and under HIP-graph replay the kernel was observed polling
// granules the memset node had not cleared yet (stale or foreign words with a non-zero tag -> wrong statistics, NaN).  The state block
// is zero-filled ONCE by the caller and must stay dedicated to one layer.
#include <algorithm>
#include "ms_common.h"
#include "maxstyle_hip.h"

namespace ms {

constexpr unsigned kSpinLimit = 1u << 22;

typedef unsigned long long u64;

// granule = {tag (launch epoch) : 32 | float bits : 32}; unit u owns granules 2u (mean) and 2u+1 (M2)
__device__ __forceinline__ void publish_granule(u64* slot, float value, unsigned tag) {
  __hip_atomic_store(slot, ((u64)tag << 32) | (u64)__float_as_uint(value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool poll_granule(const u64* slot, float& value, unsigned tag, int* err) {
  for (unsigned spins = 0;; ++spins) {
    const u64 v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(v >> 32) == tag) { value = __uint_as_float((unsigned)(v & 0xFFFFFFFFull)); return true; }
    if (spins > kSpinLimit) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); value = 0.f; return false; }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int NV, int kFusedThreads>
__global__ __launch_bounds__(kFusedThreads) void style_fused_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mu,
                                                                   float* __restrict__ sig, float* __restrict__ gamma_std, float* __restrict__ beta_std,
                                                                   int compute_std, const float* __restrict__ lmda, const float* __restrict__ gamma_noise,
                                                                   const float* __restrict__ beta_noise, const int64_t* __restrict__ perm,
                                                                   float* __restrict__ coefA, float* __restrict__ coefS, u64* __restrict__ part,
                                                                   int* __restrict__ arrive, int* __restrict__ counter, int* __restrict__ err,
                                                                   int B, int C, int HW, int S, int chunk, float eps) {
  __shared__ float red[16];
  __shared__ double redd[16];
  __shared__ float smu[256], ssig[256];
  __shared__ float pmean[1024], pm2[1024];          // polled partials of the channel: B*S <= 512 units (host-checked)
  __shared__ unsigned s_tag;
  const int tid = threadIdx.x;
  const int G = B * S, total = C * G;
  // `counter` is the epoch word, `arrive` counts finished workgroups of this launch
  if (tid == 0) {
    unsigned t = (unsigned)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    s_tag = (t == 0u) ? 1u : t;                     // 0 is the zero-filled (never published) state
  }
  __syncthreads();
  const unsigned tag = s_tag;
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int c = t / G, r = t - c * G, b = r / S, s = r - b * S;
    const int p = b * C + c;
    const int beg = s * chunk, end = min(HW, beg + chunk);
    const float* xp = x + (size_t)p * HW;
    float4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = beg + (j * kFusedThreads + tid) * 4;
      v[j] = (i < end) ? *reinterpret_cast<const float4*>(xp + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    const float n = (float)(end - beg);
    const float mean_c = block_sum(sum, red) / n;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = beg + (j * kFusedThreads + tid) * 4;
      if (i < end) {
        const float d0 = v[j].x - mean_c, d1 = v[j].y - mean_c, d2 = v[j].z - mean_c, d3 = v[j].w - mean_c;
        m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
    m2 = block_sum(m2, red);
    u64* gran = part + 2 * ((size_t)c * G);          // granules of channel c: unit (bb, ss) -> index 2*(bb*S+ss) (+1)
    if (tid == 0) { publish_granule(gran + 2 * (b * S + s), mean_c, tag); publish_granule(gran + 2 * (b * S + s) + 1, m2, tag); }
    // one thread per granule polls until it is published (our own two come back from memory as well)
    for (int q = tid; q < 2 * G; q += kFusedThreads) {
      float val;
      poll_granule(gran + q, val, tag, err);
      if (q & 1) pm2[q >> 1] = val; else pmean[q >> 1] = val;
    }
    __syncthreads();
    // every plane of channel c: Chan-merge its S chunk partials (chunk sizes follow from the geometry)
    for (int bb = tid; bb < B; bb += kFusedThreads) {
      double nn = 0.0, mean = 0.0, mm2 = 0.0;
      for (int ss = 0; ss < S; ++ss) {
        const float pm = pmean[bb * S + ss], pq = pm2[bb * S + ss];
        const double cn = (double)(min(HW, (ss + 1) * chunk) - ss * chunk);
        chan_merge(nn, mean, mm2, cn, (double)pm, (double)pq);
      }
      smu[bb] = (float)mean;
      ssig[bb] = sqrtf((float)(mm2 / (double)(HW - 1)) + eps);
    }
    __syncthreads();
    float gs, bs;
    if (compute_std & 1) {
      double am = 0.0, as = 0.0;
      for (int bb = tid; bb < B; bb += kFusedThreads) { am += (double)smu[bb]; as += (double)ssig[bb]; }
      const double mean_mu = block_sum_d(am, redd) / B;
      const double mean_sg = block_sum_d(as, redd) / B;
      double qm = 0.0, qs = 0.0;
      for (int bb = tid; bb < B; bb += kFusedThreads) {
        const double d1 = (double)smu[bb] - mean_mu, d2 = (double)ssig[bb] - mean_sg;
        qm += d1 * d1; qs += d2 * d2;
      }
      qm = block_sum_d(qm, redd); qs = block_sum_d(qs, redd);
      bs = (float)sqrt(qm / (double)(B - 1));
      gs = (float)sqrt(qs / (double)(B - 1));
    } else {
      gs = gamma_std[c]; bs = beta_std[c];
    }
    const float m = smu[b], sg = ssig[b];
    float A = sg, Sh = m;
    if (lmda != nullptr) {
      const float lam = (compute_std & 2) ? lmda[b] : fminf(fmaxf(lmda[b], 0.f), 1.f);     // bit 1: MixStyle (no clamp)
      const int pb = (int)perm[b];
      A = sg * (1.f - lam) + ssig[pb] * lam;
      Sh = m * (1.f - lam) + smu[pb] * lam;
    }
    if (gamma_noise != nullptr) {
      A += gamma_noise[p] * gs;
      Sh += beta_noise[p] * bs;
    }
    if (tid == 0 && s == 0) {
      mu[p] = m; sig[p] = sg; coefA[p] = A; coefS[p] = Sh;
      if (b == 0 && (compute_std & 1)) { gamma_std[c] = gs; beta_std[c] = bs; }
    }
    const float a = A / sg;
    float* yp = y + (size_t)p * HW;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = beg + (j * kFusedThreads + tid) * 4;
      if (i < end) {
        float4 o;
        o.x = a * (v[j].x - m) + Sh; o.y = a * (v[j].y - m) + Sh; o.z = a * (v[j].z - m) + Sh; o.w = a * (v[j].w - m) + Sh;
        *reinterpret_cast<float4*>(yp + i) = o;
      }
    }
    __syncthreads();      // smu/ssig are reused by the next unit
  }
  // end of launch: the last workgroup to get here advances the epoch (every workgroup has read it by then) and re-arms the counter
  if (tid == 0) {
    const int prev = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (int)gridDim.x - 1) {
      __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(counter, (int)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct FusedPlan { bool ok; int threads, nv, chunk, S, grid; size_t part_off, bytes; };

// State block: ints [0] epoch, [1] error word, [2] finished-workgroup counter; granules (8 B each) from byte 16.
static FusedPlan fused_plan(int B, int C, int HW) {
  FusedPlan pl{};
  pl.ok = false;
  if (HW % 4 != 0 || B < 2 || B > 256) return pl;
  // Few, fat workgroups keep the ticket counter cold (one device-wide atomic per unit, ~88 dequeues/us on one word):
  // 1024 threads x up to 64 floats hold 256 KB of a plane; fall back to 256-thread units when that leaves CUs idle.
  int threads = 1024, nv = 16;
  int S = cdiv(HW, threads * 4 * nv);
  if ((long)B * C * S >= 192) {
    while (nv > 1 && threads * 4 * (nv / 2) * S >= HW) nv >>= 1;
  } else {
    threads = 256; nv = 16;
    while (nv > 1 && (long)B * C * cdiv(HW, threads * 4 * nv) < 1024) nv >>= 1;
    while (nv < 16 && B * cdiv(HW, threads * 4 * nv) > 512) nv <<= 1;      // one channel group must fit the resident grid
    S = cdiv(HW, threads * 4 * nv);
  }
  pl.threads = threads; pl.nv = nv; pl.chunk = threads * 4 * nv; pl.S = cdiv(HW, pl.chunk);
  const int G = B * pl.S;
  // workgroups that can certainly run together: one 1024-thread (or two 256-thread) workgroup(s) per CU at <= 96 VGPRs
  const int capacity = (threads == 1024) ? 256 : 512;
  if (G > capacity) return pl;
  pl.grid = (int)std::min<long>((long)C * G, threads == 1024 ? 256L : 1024L);   // never more than fit the chip together
  if (pl.grid < G) return pl;
  pl.part_off = 16;                                                     // [0] epoch, [1] error word, [2] arrivals
  pl.bytes = pl.part_off + 2 * (size_t)C * B * pl.S * sizeof(u64);      // two tagged granules per unit
  pl.ok = true;
  return pl;
}

}  // namespace ms

using namespace ms;

extern "C" size_t ms_style_fused_ws_bytes(int B, int C, int HW) {
  const FusedPlan pl = fused_plan(B, C, HW);
  return pl.ok ? pl.bytes : 0;
}

extern "C" int ms_style_fwd_fused(const float* x, float* y, float* mu, float* sig, float* gamma_std, float* beta_std, int compute_std,
                                  const float* lmda, const float* gamma_noise, const float* beta_noise, const int64_t* perm,
                                  float* coefA, float* coefS, int B, int C, int HW, float eps, void* ws, size_t ws_bytes, void* stream) {
  const FusedPlan pl = fused_plan(B, C, HW);
  if (!pl.ok) { set_error("ms_style_fwd_fused: shape B=%d C=%d HW=%d is not eligible (use ms_style_fwd)", B, C, HW); return MS_ERR_INVALID; }
  if (!aligned16(x) || !aligned16(y)) { set_error("ms_style_fwd_fused: x and y must be 16-byte aligned"); return MS_ERR_ALIGN; }
  if (ws == nullptr || ws_bytes < pl.bytes || !aligned16(ws)) { set_error("ms_style_fwd_fused: workspace too small or unaligned"); return MS_ERR_WORKSPACE; }
  if (lmda != nullptr && perm == nullptr) { set_error("ms_style_fwd_fused: mixing needs perm"); return MS_ERR_INVALID; }
  if ((gamma_noise == nullptr) != (beta_noise == nullptr)) { set_error("ms_style_fwd_fused: gamma/beta noise must both be given"); return MS_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  int* hdr = (int*)ws;
  u64* part = (u64*)((char*)ws + pl.part_off);
  dim3 grid(pl.grid), block(pl.threads);
#define MS_FUSED(NVV, TT) MS_LAUNCH((style_fused_kernel<NVV, TT>), grid, block, 0, st, x, y, mu, sig, gamma_std, beta_std, compute_std, lmda, gamma_noise, \
                                    beta_noise, perm, coefA, coefS, part, hdr + 2, hdr, hdr + 1, B, C, HW, pl.S, pl.chunk, eps)
  if (pl.threads == 1024) {
    switch (pl.nv) { case 16: MS_FUSED(16, 1024); break; case 8: MS_FUSED(8, 1024); break; case 4: MS_FUSED(4, 1024); break;
                     case 2: MS_FUSED(2, 1024); break; default: MS_FUSED(1, 1024); break; }
  } else {
    switch (pl.nv) { case 16: MS_FUSED(16, 256); break; case 8: MS_FUSED(8, 256); break; case 4: MS_FUSED(4, 256); break;
                     case 2: MS_FUSED(2, 256); break; default: MS_FUSED(1, 256); break; }
  }
#undef MS_FUSED
  return check_launch("style_fused");
}
