// Implicit-GEMM convolution kernel template for gfx950 (exact-fp32 matrix cores, v_mfma_f32_16x16x4_f32).
// See ms_conv.hip for what it replaces in the reference and for the dispatch; this header is included by the
// translation units that instantiate slices of the template (parallel compilation).
//
// Persistent, wave-specialised workgroups (512 threads = 8 waves, 2 per SIMD):
//   waves 4-7  PRODUCERS  global --16-B loads of an aligned window--> registers --BatchNorm apply / BatchNorm-backward
//                         prologue--> LDS buffer (double buffered); they run one K-chunk ahead of the consumers
//   waves 0-3  CONSUMERS  LDS --ds_read_b32 fragments (immediate offsets)--> MFMA 16x16x4; epilogue from registers:
//                         +bias, per-wave BatchNorm statistics (count, mean, M2), 16-B stores
// A work item is (image n, output tile 8x32 or 4x16 pixels, block of 16*NT output channels); a workgroup walks items
// blockIdx.x, +gridDim.x, ... over the flattened (item, K-chunk) sequence with ONE workgroup barrier per chunk.
// Why: measured on MI355X, a conventional "every wave stages, then every wave multiplies" loop left the matrix pipe 35 %
// busy - the two waves of a SIMD run the same phase at the same time, so address arithmetic, prologue math and the epilogue
// never overlap the MFMAs.  With roles split, each SIMD holds a VALU/memory wave and an MFMA wave, which do overlap
// (separate pipes), and every global load has a full MFMA phase to land.
#pragma once
#include <map>
#include <mutex>
#include <utility>
#include <algorithm>
#include <mutex>
#include <type_traits>
#include "ms_common.h"

namespace ms {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { FETCH_NORMAL = 0, FETCH_UPS2 = 1, FETCH_ZINS2 = 2 };
#ifndef MS_CONV_PF2
#define MS_CONV_PF2 1            // 0: build the staging waves with one register set everywhere (A/B)
#endif

typedef unsigned long long conv_u64_t;
// Thread numbering of the 512-thread conv workgroups (MFMA waves = logical waves 0-3, staging waves = 4-7).  The hardware places wave h of a workgroup on SIMD
// (h + const) mod 4, so with MS_ROLE_SPLIT = 0 every SIMD hosts one MFMA wave and one staging wave of a workgroup; MS_ROLE_SPLIT = 1 makes hardware waves {0,1,4,5}
// the MFMA waves and {2,3,6,7} the staging waves: two SIMDs issue MFMAs, two stage (an fp32 MFMA blocks its SIMD's vector issue: profiles/r03_experiments.txt 15).
#ifndef MS_ROLE_SPLIT
#define MS_ROLE_SPLIT 0
#endif
#if MS_ROLE_SPLIT
__device__ __forceinline__ int conv_tid() {
  const int t = (int)threadIdx.x, h = t >> 6;
  return ((((h & 2) << 1) | (h & 1) | ((h >> 2) << 1)) << 6) | (t & 63);
}
#define MS_TID conv_tid()
#else
#define MS_TID ((int)threadIdx.x)
#endif
#ifndef MS_CONV_K1N_CK
#define MS_CONV_K1N_CK 64      // input channels per chunk of the 1x1 convs on 4x16-pixel tiles: 2 chunk barriers instead of 8 at 128 channels; a 1x1 conv accumulates its channels in the same
                              // order whatever the chunk, so every output and statistic keeps its bits (tools/k1_check.py on two builds; 16 until round 3; profiles/r03_experiments.txt 20)
#endif
#ifndef MS_CONV_K1W_CK
#define MS_CONV_K1W_CK 16      // ... of the 1x1 convs on 8x32-pixel tiles with a 16-channel output tile
#endif
#ifndef MS_CONV_K1W2_CK
#define MS_CONV_K1W2_CK 8      // ... with a 32-channel output tile (the ConvTranspose2d GEMMs)
#endif
#ifndef MS_CONV_K1W4_CK
#define MS_CONV_K1W4_CK 8      // ... with a 64-channel output tile (the 1x1 convs of config 4)
#endif
#ifndef MS_CONV_S2_PF2
#define MS_CONV_S2_PF2 0      // two register sets in the staging waves of the vector-staged stride-2 kernels with a 16-channel tile (fits with MS_CONV_S2_CK = 4: 98 VGPRs; +0.25 %, not adopted: profiles/r03_experiments.txt 17)
#endif
#ifndef MS_CONV_S2_CK
#define MS_CONV_S2_CK 8        // input channels per chunk of the stride-2 kernels with a 16-channel tile (see Geo::CK)
#endif
struct ConvArgs {
  const float* in; const float* in2; float* out; const float* w; const float* bias;
  const float* pro_a; const float* pro_b; const float* pro_c;
  float* stats;                 // float4 [cout][nparts] (count, mean, M2, 0) or null
  int N, Cin, Hs, Ws, Hin, Win, Cout, Hout, Wout, cin_pad, cout_pad;
  int pro_mode, pro_nstride, pro_cstride; float slope;
  int epi_mode, tiles_x, tiles_y, cout_real, ncb;   // ncb: number of output-channel blocks (of 16*NT)
  int wino_ok;                  // the caller accepts the Winograd form of a 3x3 stride-1 convolution for this call (MS_FETCH_WINOGRAD)
  int wino_nt1;                 // Winograd form: one channel block per staged tile (MS_FETCH_WINO_NT1)
  int wino_blocks;              // Winograd form: force the block form (MS_FETCH_WINO_BLOCKS)
  const float* wu;              // Winograd appendix of the packed weights (MS_FETCH_WINO_U): transformed weights [cb16][chunk][16][8][16], or null
  int act_bf16;                 // activation tensors (in, in2, out, mk_u) are stored as bf16 (the `_bf16` entry points); statistics / coefficients / weights fp32
  int dbg;                      // timing-only ablation bits (MS_CONV_DBG): 1 skip MFMA loop, 2 skip global loads, 4 skip epilogue stores, 8 skip LDS stores, 16 skip the epilogue
  long long* trace;             // MS_CONV_TRACE_BUILD only: cycle stamps of workgroup 0 (tools/trace_conv.py)
  // epi_mode 4 / 5 (ms_conv1x1_bnres): residual-block tail, out = lrelu(sc*mk_u + sh + conv); 5 = the conv ran at half resolution (see the epilogue)
  // epi_mode 3 (ms_conv2d_actbwd): the output is the gradient w.r.t. an activation lrelu(sc*u + sh) that was never materialised; the epilogue
  // applies its derivative and accumulates the BatchNorm-backward sums (sum g, sum g*(u - mean)) of u's layer: what ms_act_bwd_reduce does in its own pass
  const float* mk_u; const float* mk_coef; float mk_slope; float* mk_tab;   // u [N,Cout,Hout,Wout]; coef float4 [Cout] {sc,sh,mean,invstd}; tab float2 [1 + Cout*kStatSlots]
  // "cross-workgroup finalize" (the `_xfin` entry points): this launch CONSUMES BatchNorm coefficients (prologue pro_mode 1, or the residual-tail epilogue
  // epi_mode 4 / 5) that no ms_bn_finalize launch has produced yet.  xf_tab = the statistics table the producing conv wrote (its header carries a launch
  // epoch, bumped by conv_table_tail); one MFMA wave per channel - wave w of workgroup vb takes channel 4*vb + w - runs ms_bn_finalize's arithmetic on it,
  // stores the record to xf_coef (for later kernels) and publishes {tag = epoch | scale}, {tag | shift} as two 8-byte granules in xf_gran with agent-scope
  // stores; every consumer of a channel polls its granules (bounded spin; *xf_err on time-out).  See xfin_* below.
  const float* xf_tab; const float* xf_gamma; const float* xf_beta; float xf_eps; float* xf_coef; conv_u64_t* xf_gran; int* xf_err; int xf_C;
  // "rider" (ms_conv2d_ride): a ms_bn_bwd_coefs job this launch carries for the NEXT launch - MFMA wave w of workgroup b reduces channel c = 4b + w < ride_C:
  // BatchNorm-backward partial sums (ride_part [ride_C][ride_nparts] float2, or a conv-epilogue table when ride_nparts == 0) with ms_bn_bwd_coefs' arithmetic in
  // its order (same bits) and writes ride_out[c] = {al, be, de, 0} while the staging waves fetch the first chunk.  This launch itself does not read ride_out.
  // ride_kind 1: a ms_bn_finalize job instead (ride_part = the statistics table of the producing conv, ride_coef = gamma, ride_beta, ride_eps).
  const float2* ride_part; int ride_nparts; const float4* ride_coef; double ride_count; float4* ride_out; int ride_C;
  int ride_kind; const float* ride_beta; float ride_eps;
  int xf_kind; double xf_count;     // kind 0: BatchNorm statistics -> {scale, shift, mean, invstd}; kind 1: BatchNorm-backward sums (+ xf_gamma = forward records, xf_count = N*H*W) -> {al, be, de, 0}
};

typedef unsigned long long conv_u64;
// 16-byte / 8-byte table slots crossing workgroups inside one launch: 8-byte agent-scope atomics on both sides (MI355X_MICROARCH.md "Valid forms")
__device__ inline void slot_store16(float4* p, float4 v) {
  conv_u64* q = reinterpret_cast<conv_u64*>(p);
  __hip_atomic_store(q, ((conv_u64)__float_as_uint(v.y) << 32) | (conv_u64)__float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, ((conv_u64)__float_as_uint(v.w) << 32) | (conv_u64)__float_as_uint(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline float4 slot_load16(const float4* p) {
  const conv_u64* q = reinterpret_cast<const conv_u64*>(p);
  const conv_u64 lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)(lo & 0xFFFFFFFFull)), __uint_as_float((unsigned)(lo >> 32)),
                     __uint_as_float((unsigned)(hi & 0xFFFFFFFFull)), __uint_as_float((unsigned)(hi >> 32)));
}
__device__ inline void slot_store8(float2* p, float2 v) {
  __hip_atomic_store(reinterpret_cast<conv_u64*>(p), ((conv_u64)__float_as_uint(v.y) << 32) | (conv_u64)__float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline float2 slot_load8(const float2* p) {
  const conv_u64 v = __hip_atomic_load(reinterpret_cast<const conv_u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)(v & 0xFFFFFFFFull)), __uint_as_float((unsigned)(v >> 32)));
}
__device__ inline double shfl_xor_d(double v, int off) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, off, 64); hi = __shfl_xor(hi, off, 64);
  return __hiloint2double(hi, lo);
}

// ---- cross-workgroup finalize (see ConvArgs::xf_*) ------------------------------------------------------------------------------------------------
constexpr unsigned kXfinSpin = 1u << 18;
constexpr int kXfinRep = 64;        // replicas of every granule (one per lane of the publishing wave: one coalesced store); a reader takes replica (workgroup & 63)
constexpr int kXfinNG = 4;          // granules per channel: kind 0 {scale, shift, -, -}, kind 1 {al, be, de, -}
__device__ inline double xf_wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += shfl_xor_d(v, off);
  return v;
}
// the rider job (ConvArgs::ride_*): exactly bn_bwd_coefs_kernel (ms_elem.hip) run by one wave
__device__ inline void conv_ride(const ConvArgs& a, int c, int lane) {
  if (a.ride_kind == 1) {                     // exactly bn_finalize_kernel (ms_conv.hip)
    const float4* tab = reinterpret_cast<const float4*>(a.ride_part);
    const int np = (int)tab[0].x;
    const float4* part = tab + 1 + (size_t)c * kStatSlots;
    double sn = 0.0, sm = 0.0, sq = 0.0;
    for (int i = lane; i < np; i += 64) {
      const float4 q = part[i];
      const double n = (double)q.x, mu = (double)q.y;
      sn += n; sm += n * mu; sq += (double)q.z + n * mu * mu;
    }
    sn = wave_sum_d(sn); sm = wave_sum_d(sm); sq = wave_sum_d(sq);
    if (lane == 0) {
      const double mean = sm / sn;
      const double var = fmax((sq - sm * mean) / sn, 0.0);
      const float invstd = (float)(1.0 / sqrt(var + (double)a.ride_eps));
      const float sc = reinterpret_cast<const float*>(a.ride_coef)[c] * invstd;
      a.ride_out[c] = make_float4(sc, a.ride_beta[c] - (float)mean * sc, (float)mean, invstd);
    }
    return;
  }
  int nparts = a.ride_nparts;
  const float2* row = (nparts == 0) ? a.ride_part + 1 + (size_t)c * kStatSlots : a.ride_part + (size_t)c * nparts;
  if (nparts == 0) nparts = (int)a.ride_part[0].x;
  double s1 = 0.0, s2 = 0.0;
  for (int i = lane; i < nparts; i += 64) { const float2 q = row[i]; s1 += (double)q.x; s2 += (double)q.y; }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (lane == 0) {
    const float4 cf = a.ride_coef[c];           // {sc, sh, mean, invstd}
    const double mean = cf.z, invstd = cf.w, sc = cf.x;
    const double c1 = s1 / a.ride_count;
    const double c2 = s2 * invstd / a.ride_count;
    const double be = -sc * c2 * invstd;
    a.ride_out[c] = make_float4((float)sc, (float)be, (float)(-sc * c1 - be * mean), 0.f);
  }
}
__device__ inline conv_u64_t* xfin_slot(const ConvArgs& a, int c, int g, int rep) { return a.xf_gran + ((size_t)c * kXfinNG + g) * kXfinRep + rep; }
// header of the producing conv's table: {slots in use, launch epoch}; float4 records (statistics, kind 0) or float2 records (BatchNorm-backward sums, kind 1)
__device__ inline void xfin_header(const ConvArgs& a, unsigned& tag, int& nparts) {
  const float2 h = *reinterpret_cast<const float2*>(a.xf_tab);
  nparts = (int)h.x; tag = __float_as_uint(h.y);
}
__device__ inline void xfin_publish(const ConvArgs& a, int c, unsigned tag, float v0, float v1, float v2, int nv) {
  // (first version: ONE copy per granule - at 16 channels every lane of 512 workgroups polled the same two cache lines; the replicas cost one coalesced
  //  512-byte store per granule; profiles/r03_experiments.txt)
  const int lane = MS_TID & 63;
  v0 = __shfl(v0, 0, 64); v1 = __shfl(v1, 0, 64); v2 = __shfl(v2, 0, 64);
  __hip_atomic_store(xfin_slot(a, c, 0, lane), ((conv_u64_t)tag << 32) | (conv_u64_t)__float_as_uint(v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(xfin_slot(a, c, 1, lane), ((conv_u64_t)tag << 32) | (conv_u64_t)__float_as_uint(v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (nv > 2) __hip_atomic_store(xfin_slot(a, c, 2, lane), ((conv_u64_t)tag << 32) | (conv_u64_t)__float_as_uint(v2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One full wave, channel c.  kind 0: bn_finalize_kernel's arithmetic (ms_conv.hip); kind 1: bn_bwd_coefs_kernel's (ms_elem.hip) - same order, the same bits.
// The record goes to xf_coef (for later kernels), the values a consumer of THIS launch needs are published.
__device__ inline void xfin_reduce_publish(const ConvArgs& a, int c, unsigned tag, int nparts) {
  const int lane = MS_TID & 63;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f;
  if (a.xf_kind == 0) {
    const float4* part = reinterpret_cast<const float4*>(a.xf_tab) + 1 + (size_t)c * kStatSlots;
    double sn = 0.0, sm = 0.0, sq = 0.0;
    for (int i = lane; i < nparts; i += 64) {
      const float4 q = part[i];
      const double n = (double)q.x, mu = (double)q.y;
      sn += n; sm += n * mu; sq += (double)q.z + n * mu * mu;
    }
    sn = xf_wave_sum_d(sn); sm = xf_wave_sum_d(sm); sq = xf_wave_sum_d(sq);
    if (lane == 0) {
      const double mean = sm / sn;
      const double var = fmax((sq - sm * mean) / sn, 0.0);
      const float invstd = (float)(1.0 / sqrt(var + (double)a.xf_eps));
      const float sc = a.xf_gamma[c] * invstd;
      const float sh = a.xf_beta[c] - (float)mean * sc;
      reinterpret_cast<float4*>(a.xf_coef)[c] = make_float4(sc, sh, (float)mean, invstd);
      p0 = sc; p1 = sh;
    }
    xfin_publish(a, c, tag, p0, p1, 0.f, 2);
  } else {
    const float2* row = reinterpret_cast<const float2*>(a.xf_tab) + 1 + (size_t)c * kStatSlots;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < nparts; i += 64) { const float2 q = row[i]; s1 += (double)q.x; s2 += (double)q.y; }
    s1 = xf_wave_sum_d(s1); s2 = xf_wave_sum_d(s2);
    if (lane == 0) {
      const float4 cf = reinterpret_cast<const float4*>(a.xf_gamma)[c];           // kind 1: xf_gamma = the forward records {sc, sh, mean, invstd}
      const double mean = cf.z, invstd = cf.w, sc = cf.x;
      const double c1 = s1 / a.xf_count;
      const double c2 = s2 * invstd / a.xf_count;
      const double be = -sc * c2 * invstd;
      p0 = (float)sc; p1 = (float)be; p2 = (float)(-sc * c1 - be * mean);
      reinterpret_cast<float4*>(a.xf_coef)[c] = make_float4(p0, p1, p2, 0.f);
    }
    xfin_publish(a, c, tag, p0, p1, p2, 3);
  }
}
// The MFMA waves of the launch share the channels out: wave w of workgroup b (HARDWARE block index, not the XCD-permuted vb) reduces channels 4*b + w, ...
// Workgroups are dispatched in block-index order, so the publishers are the first ones on the chip: a resident workgroup never waits for one that is
// not - also when the grid does not fit the chip at once (found by the bounded spin: the 64-channel-tile 1x1 kernel needs 205 VGPRs, one workgroup per
// CU, and with publishers numbered by vb half of them sat behind their own pollers).  Returns whether this wave published anything.
__device__ inline bool xfin_produce(const ConvArgs& a, unsigned tag, int nparts) {
  const int wave = MS_TID >> 6;
  bool any = false;
  for (int c = 4 * (int)blockIdx.x + wave; c < a.xf_C; c += 4 * (int)gridDim.x) { xfin_reduce_publish(a, c, tag, nparts); any = true; }
  return any;
}
// both granules of a channel per round trip (each validates itself by its tag, so a torn pair is simply retried); rep = the workgroup's replica
__device__ inline void xfin_peek(const ConvArgs& a, int c, int rep, conv_u64_t (&g)[2]) {
  g[0] = __hip_atomic_load(xfin_slot(a, c, 0, rep), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  g[1] = __hip_atomic_load(xfin_slot(a, c, 1, rep), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline float2 xfin_poll(const ConvArgs& a, int c, int rep, unsigned tag, conv_u64_t (&g)[2]) {      // g: an earlier peek (issued long before this call)
  for (unsigned spins = 0;; ++spins) {
    if ((unsigned)(g[0] >> 32) == tag && (unsigned)(g[1] >> 32) == tag)
      return make_float2(__uint_as_float((unsigned)(g[0] & 0xFFFFFFFFull)), __uint_as_float((unsigned)(g[1] & 0xFFFFFFFFull)));
    if (spins > kXfinSpin) { __hip_atomic_store(a.xf_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return make_float2(0.f, 0.f); }
    __builtin_amdgcn_s_sleep(8);
    xfin_peek(a, c, rep, g);
  }
}
// PROLOGUE consumers (pro_mode 1 / 2 with xf_tab): `nthreads` threads (the MFMA waves, tid = 0..nthreads-1) fill the launch's LDS coefficient table - one float4
// {a, b, c, 0} per input channel, `ntab` entries, (d0, d1, 0) beyond Cin - from the published granules: NV = 2 (BatchNorm apply) or 3 (BatchNorm backward).
template <int NV>
__device__ inline void xfin_fill(const ConvArgs& a, float* cf_lds, int ntab, unsigned tag, int rep, int tid, int nthreads, float d0, float d1) {
  for (int c = tid; c < ntab; c += nthreads) {
    float v[3] = {d0, d1, 0.f};
    if (c < a.Cin) {
      conv_u64_t g[3];
      for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NV; ++i) { g[i] = __hip_atomic_load(xfin_slot(a, c, i, rep), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#pragma unroll
        for (int i = 0; i < NV; ++i) ok = ok && ((unsigned)(g[i] >> 32) == tag);
        if (ok) break;
        if (spins > kXfinSpin) { __hip_atomic_store(a.xf_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); g[0] = g[1] = g[2] = 0; break; }
        __builtin_amdgcn_s_sleep(8);
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = __uint_as_float((unsigned)(g[i] & 0xFFFFFFFFull));
    }
    reinterpret_cast<float4*>(cf_lds)[c] = make_float4(v[0], v[1], v[2], 0.f);
  }
}

// End of a conv kernel, MFMA waves only (threads 0..255; the staging waves have returned - s_barrier only waits for surviving waves): the per-lane
// partials of the epilogue become ONE table slot per workgroup and channel (table[1 + co*kStatSlots + workgroup-within-channel-block], [0] = slots in use),
//   STATS:  v0 = count, v1[j] = mean, v2[j] = M2 (per lane)  -> float4 slots {n, mean, M2, 0}
//   !STATS: v1[j] = sum g, v2[j] = sum g*(u - mean) (per lane) -> float2 slots
template <int NT, bool STATS>
__device__ inline void conv_table_tail(const ConvArgs& a, float* smem, int vb, int ncb, float v0, const float (&v1)[NT], const float (&v2)[NT]) {
  constexpr int COUT_TILE = 16 * NT;
  const int lane = MS_TID & 63, wave = MS_TID >> 6, m = lane & 15, k = lane >> 4;
  const int cb0 = vb % ncb, wg = vb / ncb, S = (int)gridDim.x / ncb;
  auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  bar();                                               // every MFMA wave is done with the stage buffers: smem is free
  float* red = smem;                                   // [4 waves][COUT_TILE][3]
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float n_ = v0, a_ = v1[j], b_ = v2[j];
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {         // the four lanes (k = 0..3) that hold the same channel
      const float nb = __shfl_xor(n_, off, 64), ab = __shfl_xor(a_, off, 64), bb = __shfl_xor(b_, off, 64);
      if (STATS) {
        const float nn = n_ + nb;
        const float w = (nn > 0.f) ? nb / nn : 0.f;
        const float d = ab - a_;
        a_ += d * w; b_ += bb + d * d * n_ * w; n_ = nn;
      } else { a_ += ab; b_ += bb; }
    }
    if (k == 0) { float* r = red + ((wave * COUT_TILE) + j * 16 + m) * 3; r[0] = n_; r[1] = a_; r[2] = b_; }
  }
  bar();
  if (MS_TID < COUT_TILE) {
    const int ch = MS_TID;
    float n_ = red[ch * 3], a_ = red[ch * 3 + 1], b_ = red[ch * 3 + 2];
#pragma unroll
    for (int w_ = 1; w_ < 4; ++w_) {
      const float* r = red + ((w_ * COUT_TILE) + ch) * 3;
      if (STATS) {
        const float nn = n_ + r[0];
        const float w = (nn > 0.f) ? r[0] / nn : 0.f;
        const float d = r[1] - a_;
        a_ += d * w; b_ += r[2] + d * d * n_ * w; n_ = nn;
      } else { a_ += r[1]; b_ += r[2]; }
    }
    const int co = cb0 * COUT_TILE + ch;
    if (co < a.Cout) {
      if (STATS) {
        float4* slot = reinterpret_cast<float4*>(a.stats) + 1 + (size_t)co * kStatSlots + wg;
        *slot = make_float4(n_, a_, b_, 0.f);
      } else {
        float2* slot = reinterpret_cast<float2*>(a.mk_tab) + 1 + (size_t)co * kStatSlots + wg;
        *slot = make_float2(a_, b_);
      }
    }
  }
  if (vb == 0 && MS_TID == 0) {
    // header: {slots in use, launch epoch of this table (an integer in float bits: the tag of the `_xfin` consumers' granules), 0, 0}
    if (STATS) {
      const unsigned ep = __float_as_uint(reinterpret_cast<const float4*>(a.stats)[0].y) + 1u;
      reinterpret_cast<float4*>(a.stats)[0] = make_float4((float)S, __uint_as_float(ep == 0u ? 1u : ep), 0.f, 0.f);
    } else {
      const unsigned ep = __float_as_uint(reinterpret_cast<const float2*>(a.mk_tab)[0].y) + 1u;
      reinterpret_cast<float2*>(a.mk_tab)[0] = make_float2((float)S, __uint_as_float(ep == 0u ? 1u : ep));
    }
  }
}


template <int KS, int STRIDE, int FETCH, bool VEC, bool NARROW, int NT>
struct Geo {
  static constexpr int TW = NARROW ? 16 : 32;
  // narrow: 4x16 pixels, one M-tile per wave - the 16-pixel-wide layers are the small-spatial ones where work items, not operand reuse,
  // are scarce (measured at C2: 16x16 tiles 272.5 steps/s, 8x16 287.3, 4x16 290.2)
  static constexpr int TH = NARROW ? 4 : 8;
  static constexpr int MT = NARROW ? 1 : 4;                              // 16-pixel M-tiles per consumer wave
  static constexpr int PAD = (KS == 3) ? 1 : 0;
  static constexpr int PADL = VEC ? ((KS == 3) ? 4 : 0) : PAD;           // window starts PADL logical columns left of ox0*S
  static constexpr int IH = (TH - 1) * STRIDE + KS;
  static constexpr int WIN_W = VEC ? (TW * STRIDE + ((KS == 3) ? 8 : 0)) : ((TW - 1) * STRIDE + KS);
  static constexpr int HALF = (WIN_W + 1) / 2;
  static constexpr int RS = (STRIDE == 1) ? ((WIN_W + 3) / 4 * 4) : 2 * HALF;
  static constexpr int BASE = IH * RS;
  static constexpr int PS = BASE + ((16 - BASE % 32 + 32) % 32);           // plane stride == 16 (mod 32 banks)
  // EXP: the LDS tile is the LOGICAL (2x up-sampled / zero-inserted) input, staged from 8-byte loads of the STORED tensor:
  // one slot = 2 stored values = 4 logical columns (UPS2: both logical rows 2sr-1, 2sr; ZINS2: logical row 2sr+1, odd columns 0)
  static constexpr bool EXP = VEC && (FETCH != FETCH_NORMAL);
  // input channels per K-chunk: sized so that two LDS buffers of (input tile + weight slice) leave >= 2 workgroups per CU
  static constexpr int CK = (KS == 1 && NARROW && VEC && NT == 1) ? MS_CONV_K1N_CK : ((KS == 1 && !NARROW && VEC && NT == 1) ? MS_CONV_K1W_CK : ((KS == 1 && !NARROW && VEC && NT == 2) ? MS_CONV_K1W2_CK : ((KS == 1 && !NARROW && VEC && NT == 4) ? MS_CONV_K1W4_CK :
                            ((STRIDE == 1 && VEC && NT == 1) ? 16 : ((STRIDE == 2 && NT > 1) ? 4 : ((STRIDE == 2) ? MS_CONV_S2_CK : 8))))));
  static constexpr int VW = EXP ? 2 : (VEC ? 4 : 1);                       // elements per staging load
  static constexpr int SR = EXP ? ((FETCH == FETCH_UPS2) ? IH / 2 + 1 : IH / 2) : IH;   // staged rows per channel
  static constexpr int ROW_ITEMS = EXP ? WIN_W / 4 : WIN_W / VW;           // VEC: WIN_W % 4 == 0 by construction
  static constexpr int ITEMS = CK * SR * ROW_ITEMS;
  static constexpr int NI = (ITEMS + 255) / 256;
  static constexpr int WS = (NT == 1) ? 16 : NT * 16 + 16;                 // weight row stride (bank-conflict-free B fragments)
  static constexpr int TAPS = KS * KS;
  static constexpr int BUF = CK * PS + TAPS * CK * WS;                     // floats per LDS stage buffer

  static constexpr int tap_off(int tap) {                                  // LDS offset of tap (ky,kx) relative to the pixel's slot
    const int ky = tap / KS, kx = tap % KS;
    const int t = kx - PAD + PADL;
    return ky * RS + ((STRIDE == 1) ? t : ((t & 1) * HALF + (t >> 1)));
  }
};

// AT = storage type of the activation tensors in / in2 / out / mk_u (float or ms_bf16, ms_common.h ActIO): everything else is fp32 either way
template <int KS, int STRIDE, int FETCH, int NT, bool VEC, bool NARROW, bool IN2, typename AT = float>
__global__ __launch_bounds__(512, (NT == 1 ? 4 : 2)) void conv_mfma_kernel(const ConvArgs a) {
  using G = Geo<KS, STRIDE, FETCH, VEC, NARROW, NT>;
  using IO = ActIO<AT>;
  constexpr int AB = IO::kBytes;
  constexpr int CK = G::CK, PS = G::PS, RS = G::RS, IH = G::IH, PAD = G::PAD, PADL = G::PADL, HALF = G::HALF;
  constexpr int TW = G::TW, TH = G::TH, VW = G::VW, ROW_ITEMS = G::ROW_ITEMS, ITEMS = G::ITEMS, NI = G::NI, SR = G::SR;
  constexpr bool EXP = G::EXP;
  constexpr int WS = G::WS, TAPS = G::TAPS, BUF = G::BUF;
  constexpr int COUT_TILE = 16 * NT;
  constexpr int WITEMS = TAPS * CK * (COUT_TILE / 4);
  constexpr int NWI = (WITEMS + 255) / 256;
  // two register sets in the staging waves (prefetch distance 2, see StageRegs) where a set is small enough to double inside the kernel's register budget
  // (chosen per variant from `tools/kernel_resources.sh`: on only where the kernel stays within 128 VGPRs without spills - the narrow 4x16-pixel tiles,
  //  the 1x1 convolutions and the expanded-fetch 3x3: the small-spatial layers that are latency-bound, not the top-level ones that are bandwidth-bound)
  constexpr bool PF2 = MS_CONV_PF2 && ((IN2 ? 2 : 1) * NI * VW + 4 * NWI <= 56) &&
                       (NARROW ? !(NT == 4 && IN2) : ((KS == 1 && NT < 4) || (EXP && NT == 1) || (MS_CONV_S2_PF2 && STRIDE == 2 && NT == 1 && VEC)));
  static_assert(!EXP || (KS == 3 && STRIDE == 1), "expanded staging is for the 3x3 stride-1 convolution");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // smem: [2][BUF] stage buffers (input tile [CK][PS] then weights [TAPS][CK][WS]) | [cin_pad][4] prologue coefficients
  float* cf_lds = smem + 2 * BUF;

  const int wave = MS_TID >> 6, lane = MS_TID & 63;
  const bool producer = wave >= 4;                       // wave-uniform role
  const int ntiles = a.tiles_x * a.tiles_y;
  const int ncb = a.ncb;
  const int nitems = a.N * ntiles * ncb;
  const int nchunks = (a.cin_pad + CK - 1) / CK;
  // XCD-aware numbering: hardware deals workgroups round-robin over the 8 XCDs (b and b+8 share one), so give XCD x the contiguous
  // run [x*grid/8, (x+1)*grid/8) of every round - neighbouring tiles (shared halo rows/columns, shared weights) then meet in ONE L2.
  // Speed only: any placement computes the same result.
  const int vb = ((int)gridDim.x % 8 == 0) ? (((int)blockIdx.x % 8) * ((int)gridDim.x / 8) + (int)blockIdx.x / 8) : (int)blockIdx.x;
  const int my_items = (nitems - vb + (int)gridDim.x - 1) / (int)gridDim.x;   // >= 1 (grid <= nitems)
  const int T = my_items * nchunks;                       // (item, chunk) iterations of this workgroup
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // orders LDS only (no vmcnt drain)
  auto decode = [&](int it, int& n, int& tile, int& cb) { cb = it % ncb; const int t2 = it / ncb; tile = t2 % ntiles; n = t2 / ntiles; };

  if constexpr (EXP && FETCH == FETCH_ZINS2) {
    for (int i = MS_TID; i < 2 * BUF; i += 512) smem[i] = 0.f;     // rows / columns the staging never writes are the inserted zeros
  }
  // per-channel prologue coefficients are constant for the whole launch: stage them once (per-plane mode reads global memory)
  if (a.pro_mode != 0 && a.pro_nstride == 0 && a.xf_tab == nullptr) {
    for (int c = MS_TID; c < a.cin_pad; c += 512) {
      float ca = 1.f, cb_ = 0.f, cc = 0.f;
      if (c < a.Cin) { ca = a.pro_a[c * a.pro_cstride]; cb_ = a.pro_b[c * a.pro_cstride]; if (a.pro_mode == 2) cc = a.pro_c[c * a.pro_cstride]; }
      cf_lds[c * 4] = ca; cf_lds[c * 4 + 1] = cb_; cf_lds[c * 4 + 2] = cc;
    }
  }

  if (producer) {
    // =========================================== PRODUCER waves ===========================================
    __builtin_amdgcn_s_setprio(3);
    const int tid = MS_TID - 256;
    const size_t in_plane = (size_t)a.Hs * a.Ws;
    int s_lds[NI];        // (channel-in-chunk << 20) | LDS float offset of the slot, or -1: no slot   (tile independent)
    int s_rw[NI];         // (window row << 16) | window column
    int s_goff[NI];       // global offset inside one channel plane (or -1: outside the image -> zeros)  (per tile)
    // BUF_LD (16-byte loads of the stored tensor: the common path): buffer addressing, as in the wide kernel (ms_conv_wide.h) - resource = image n, scalar
    // offset = first channel of the chunk, vector offset = s_boff[j] = byte offset of (channel-in-chunk, pixel) computed ONCE PER TILE, or a sentinel beyond
    // num_records for slots outside the image: the hardware range check returns the zeros of the padding, no per-chunk address arithmetic, selects or branches
    constexpr bool BUF_LD = VEC && !EXP;
    constexpr int kOob = 0x7FFFFFF0;
    int s_boff[BUF_LD ? NI : 1];
    unsigned tile_ok = 0;  // bit j: slot j lies inside the image for the current tile
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int it = tid + j * 256;
      s_lds[j] = -1; s_rw[j] = 0; s_goff[j] = -1;
      if (it < ITEMS) {
        const int f = it % ROW_ITEMS;
        const int row = it / ROW_ITEMS;
        const int r = row % SR, c = row / SR;
        if constexpr (EXP) {
          // logical row of the slot's first LDS write: UPS2 rows (2r-1, 2r), ZINS2 row 2r+1; logical column 4f
          const int lr = (FETCH == FETCH_UPS2) ? (2 * r - 1) : (2 * r + 1);
          s_lds[j] = (c << 20) | (c * PS + (lr + 1) * RS + 4 * f);        // +1 row bias keeps the packed offset non-negative for lr = -1
          s_rw[j] = (r << 16) | f;
        } else {
          const int w = f * VW;
          const int q = (STRIDE == 1) ? w : ((w & 1) * HALF + (w >> 1));
          s_lds[j] = (c << 20) | (c * PS + r * RS + q);
          s_rw[j] = (r << 16) | w;
        }
      }
    }
    auto set_tile = [&](int tile) {
      const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
      const int iy0 = ty * TH * STRIDE - PAD;
      const int wx0 = tx * TW * STRIDE - PADL;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if constexpr (EXP) {
          // stored coordinates of the slot's two values (tile origins are even, so the shifts are exact)
          const int sr = s_rw[j] >> 16, f = s_rw[j] & 0xFFFF;
          const int Ys = ((ty * TH) >> 1) + sr - ((FETCH == FETCH_UPS2) ? 1 : 0);
          const int Xs = ((tx * TW) >> 1) - 2 + 2 * f;
          const bool ok = (s_lds[j] >= 0) && (Ys >= 0) && (Ys < a.Hs) && (Xs >= 0) && (Xs < a.Ws);
          s_goff[j] = ok ? (Ys * a.Ws + Xs) : -1;
        } else {
          const int Y = iy0 + (s_rw[j] >> 16), X = wx0 + (s_rw[j] & 0xFFFF);
          bool ok = (s_lds[j] >= 0) && (Y >= 0) && (Y < a.Hin) && (X >= 0) && (X < a.Win);
          int ys = Y, xs = X;
          if (FETCH == FETCH_UPS2) { ys = Y >> 1; xs = X >> 1; }
          if (FETCH == FETCH_ZINS2) { ok = ok && !((Y | X) & 1); ys = Y >> 1; xs = X >> 1; }
          s_goff[j] = ok ? (ys * a.Ws + xs) : -1;
        }
      }
      if constexpr (BUF_LD) {
        tile_ok = 0;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const bool ok = (s_lds[j] >= 0) && (s_goff[j] >= 0);
          tile_ok |= (ok ? 1u : 0u) << j;
          s_boff[j] = ok ? AB * ((s_lds[j] >> 20) * (int)in_plane + s_goff[j]) : kOob;
        }
      }
    };
    // One chunk in flight between its global loads and its LDS stores.  With G::PF2 the staging waves keep TWO such sets and run two chunks ahead of the
    // MFMA waves (the loop below is unrolled by two, one set per half): a chunk's loads then have two chunk periods to land instead of one.  The small-spatial
    // deep layers (128 channels at 16x16: 8 chunks of ~0.5 us of matrix work each, one item per workgroup) were bound by exactly that latency.
    struct StageRegs {
      float rin[NI][VW];
      float rin2[IN2 ? NI : 1][VW];
      float4 rw[NWI];
      unsigned okmask;              // bit j: slot j of the chunk held in registers lies inside the image
      bool have_w;                  // rw holds a weight slice that must be written to LDS
      int n, c0;                    // image and first input channel of the chunk
    };

    typedef unsigned lu32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned lu32x2_t __attribute__((ext_vector_type(2)));
    auto load_chunk = [&](StageRegs& R, int n, int co0, int c0, bool load_w) {
      R.n = n; R.c0 = c0;
      const size_t img_off = (size_t)n * a.Cin * in_plane;                        // element offset of image n (both input tensors)
      if constexpr (BUF_LD) {
        const bool ragged = (c0 + CK > a.Cin);
        R.okmask = tile_ok;
        const unsigned img_bytes = (unsigned)AB * (unsigned)a.Cin * (unsigned)in_plane;          // the host checks Cin*plane*4 < 2^31
        char* in_n = const_cast<char*>(reinterpret_cast<const char*>(a.in)) + img_off * AB;
        char* in2_n = IN2 ? const_cast<char*>(reinterpret_cast<const char*>(a.in2)) + img_off * AB : in_n;
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(in_n, 0, img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(in2_n, 0, img_bytes, 0x00020000);
        const int soff = AB * c0 * (int)in_plane;
        const bool dead = (a.dbg & 2) != 0;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          int off = s_boff[j];
          if (ragged || dead) {                       // (wave-uniform) a layer's last chunk: channels beyond Cin are zeros too
            const bool ok = ((R.okmask >> j) & 1u) && (c0 + (s_lds[j] >> 20) < a.Cin) && !dead;
            if (!ok) { off = kOob; R.okmask &= ~(1u << j); }
          }
          if constexpr (AB == 4) {
            const lu32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r1, off, soff, 0);
            R.rin[j][0] = __uint_as_float(v.x); R.rin[j][1] = __uint_as_float(v.y); R.rin[j][2] = __uint_as_float(v.z); R.rin[j][3] = __uint_as_float(v.w);
            if constexpr (IN2) {
              const lu32x4_t u = __builtin_amdgcn_raw_buffer_load_b128(r2, off, soff, 0);
              R.rin2[j][0] = __uint_as_float(u.x); R.rin2[j][1] = __uint_as_float(u.y); R.rin2[j][2] = __uint_as_float(u.z); R.rin2[j][3] = __uint_as_float(u.w);
            }
          } else {                                    // bf16 storage: 4 values = one 8-byte load, widened to fp32 (a shift / a mask each)
            const lu32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r1, off, soff, 0);
            R.rin[j][0] = __uint_as_float(v.x << 16); R.rin[j][1] = __uint_as_float(v.x & 0xFFFF0000u); R.rin[j][2] = __uint_as_float(v.y << 16); R.rin[j][3] = __uint_as_float(v.y & 0xFFFF0000u);
            if constexpr (IN2) {
              const lu32x2_t u = __builtin_amdgcn_raw_buffer_load_b64(r2, off, soff, 0);
              R.rin2[j][0] = __uint_as_float(u.x << 16); R.rin2[j][1] = __uint_as_float(u.x & 0xFFFF0000u); R.rin2[j][2] = __uint_as_float(u.y << 16); R.rin2[j][3] = __uint_as_float(u.y & 0xFFFF0000u);
            }
          }
        }
      } else {
      R.okmask = 0;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int ci = c0 + (s_lds[j] >> 20);
        const bool ok = (s_lds[j] >= 0) && (s_goff[j] >= 0) && (ci < a.Cin);
        R.okmask |= (ok ? 1u : 0u) << j;
        const bool ld = ok && !(a.dbg & 2);
        const size_t off = ld ? ((size_t)ci * in_plane + (size_t)s_goff[j]) : 0;
        if constexpr (EXP) {
          const float2 v = ld ? IO::ld2(a.in, img_off + off) : make_float2(0.f, 0.f);
          R.rin[j][0] = v.x; R.rin[j][1] = v.y;
          if constexpr (IN2) {
            const float2 u = ld ? IO::ld2(a.in2, img_off + off) : make_float2(0.f, 0.f);
            R.rin2[j][0] = u.x; R.rin2[j][1] = u.y;
          }
        } else if constexpr (VEC) {
          const float4 v = ld ? IO::ld4(a.in, img_off + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          R.rin[j][0] = v.x; R.rin[j][1] = v.y; R.rin[j][2] = v.z; R.rin[j][3] = v.w;
          if constexpr (IN2) {
            const float4 u = ld ? IO::ld4(a.in2, img_off + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            R.rin2[j][0] = u.x; R.rin2[j][1] = u.y; R.rin2[j][2] = u.z; R.rin2[j][3] = u.w;
          }
        } else {
          R.rin[j][0] = ld ? IO::ld1(a.in, img_off + off) : 0.f;
          if constexpr (IN2) R.rin2[j][0] = ld ? IO::ld1(a.in2, img_off + off) : 0.f;
        }
      }
      }
      R.have_w = load_w;
      if (load_w) {
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
          const int idx = tid + j * 256;
          const int j4 = idx % (COUT_TILE / 4);
          const int row = idx / (COUT_TILE / 4);      // tap*CK + c
          const int c = row % CK, tap = row / CK;
          R.rw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx < WITEMS && c0 + c < a.cin_pad)
            R.rw[j] = *reinterpret_cast<const float4*>(a.w + ((size_t)tap * a.cin_pad + c0 + c) * a.cout_pad + co0 + j4 * 4);
        }
      }
    };

    auto store_chunk = [&](StageRegs& R, float* buf) {
      const int n = R.n, c0 = R.c0;
      float* in_lds = buf;
      float* w_lds = buf + CK * PS;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (s_lds[j] < 0) continue;
        const int ci = c0 + (s_lds[j] >> 20);
        float v[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) v[e] = R.rin[j][e];
        if ((R.okmask >> j) & 1u) {
          if (a.pro_mode == 1) {
            float pa, pb;
            if (a.pro_nstride == 0) { pa = cf_lds[ci * 4]; pb = cf_lds[ci * 4 + 1]; }
            else { const int pi = (n * a.pro_nstride + ci) * a.pro_cstride; pa = a.pro_a[pi]; pb = a.pro_b[pi]; }
#pragma unroll
            for (int e = 0; e < VW; ++e) v[e] = leaky(pa * v[e] + pb, a.slope);
          } else if constexpr (IN2) {
            if (a.pro_mode == 2) {
              const float pa = cf_lds[ci * 4], pb = cf_lds[ci * 4 + 1], pc = cf_lds[ci * 4 + 2];
#pragma unroll
              for (int e = 0; e < VW; ++e) v[e] = pa * v[e] + pb * R.rin2[j][e] + pc;
            }
          }
        }
        float* dst = in_lds + (s_lds[j] & 0xFFFFF);
        if constexpr (EXP) {
          dst -= RS;                                      // undo the +1 row bias: dst = logical row lr, column 4f
          const int sr = s_rw[j] >> 16;
          if constexpr (FETCH == FETCH_UPS2) {
            const float4 e = make_float4(v[0], v[0], v[1], v[1]);
            if (sr > 0) *reinterpret_cast<float4*>(dst) = e;                     // logical row 2sr-1
            if (2 * sr < IH) *reinterpret_cast<float4*>(dst + RS) = e;           // logical row 2sr
          } else {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], 0.f, v[1], 0.f);   // logical row 2sr+1; odd columns are inserted zeros
          }
        } else if constexpr (VEC) {
          if constexpr (STRIDE == 1) {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          } else {          // de-interleave even / odd columns (w % 4 == 0 -> even half at dst, odd half at dst+HALF)
            *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[2]);
            *reinterpret_cast<float2*>(dst + HALF) = make_float2(v[1], v[3]);
          }
        } else {
          dst[0] = v[0];
        }
      }
      if (R.have_w) {
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
          const int idx = tid + j * 256;
          if (idx < WITEMS) {
            const int j4 = idx % (COUT_TILE / 4);
            const int row = idx / (COUT_TILE / 4);
            *reinterpret_cast<float4*>(w_lds + row * WS + j4 * 4) = R.rw[j];
          }
        }
      }
    };

    // iteration p (chunk index in this workgroup's sequence): a register set holds chunk p; it is written to buffer p&1.
    // key_*[b]: which weight slice (cb, c0) buffer b holds -> skip the rewrite when it repeats (single-chunk layers)
    int key_cb[2] = {-1, -1}, key_c0[2] = {-1, -1};
    int item = vb, chunk = 0, n, tile, cb, tile_set = -1;     // the LOAD cursor
    decode(item, n, tile, cb);
    set_tile(tile); tile_set = tile;
    auto advance = [&]() {
      if (++chunk == nchunks) { chunk = 0; item += gridDim.x; decode(item, n, tile, cb); }
      if (tile != tile_set) { set_tile(tile); tile_set = tile; }
    };
    if constexpr (PF2) {
      StageRegs R0, R1;
      R0.okmask = R1.okmask = 0; R0.have_w = R1.have_w = false;
      // chunk q goes to buffer q&1, whose weight key is decided when chunk q-2 is stored: a load compares against the key its buffer WILL hold then
      auto issue = [&](StageRegs& R, int b) {
        const bool same = (key_cb[b] == cb && key_c0[b] == chunk * CK);
        load_chunk(R, n, cb * COUT_TILE, chunk * CK, !same);
        key_cb[b] = cb; key_c0[b] = chunk * CK;
      };
      issue(R0, 0);
      if (T > 1) { advance(); issue(R1, 1); }
      lds_barrier();                                  // barrier #0: coefficient table visible (matched by the consumers)
      for (int p = 0; p < T; p += 2) {
        store_chunk(R0, smem);
        if (p + 2 < T) { advance(); issue(R0, 0); }
        lds_barrier();                                // barrier #(p+1): chunk p visible; consumers done with chunk p-1
        if (p + 1 < T) {
          store_chunk(R1, smem + BUF);
          if (p + 3 < T) { advance(); issue(R1, 1); }
          lds_barrier();                              // barrier #(p+2)
        }
      }
    } else {
      StageRegs R;
      R.okmask = 0; R.have_w = false;
      load_chunk(R, n, cb * COUT_TILE, 0, true);
      lds_barrier();                                    // barrier #0: coefficient table visible (matched by the consumers)
      for (int p = 0; p < T; ++p) {
        // write chunk p (in registers) into buffer p&1; the consumers read buffer (p-1)&1 meanwhile
        store_chunk(R, smem + (p & 1) * BUF);
        key_cb[p & 1] = cb; key_c0[p & 1] = chunk * CK;
        // advance to chunk p+1 and issue its loads
        if (p + 1 < T) {
          advance();
          const int b = (p + 1) & 1;
          load_chunk(R, n, cb * COUT_TILE, chunk * CK, !(key_cb[b] == cb && key_c0[b] == chunk * CK));
        }
        lds_barrier();                                  // barrier #(p+1): chunk p visible; consumers done with chunk p-1
      }
    }
    return;
  }

  // =========================================== CONSUMER waves ===========================================
  // (hardware index: the first workgroups dispatched.  The channels are dealt round-robin over the launch's MFMA waves, so ANY grid carries the whole job - with
  //  fewer than ride_C / 4 workgroups a wave takes more than one channel: ms_conv_ride_capacity is a speed hint, no longer a correctness bound)
  if (a.ride_out != nullptr)
    for (int c = (int)blockIdx.x * 4 + wave; c < a.ride_C; c += 4 * (int)gridDim.x) conv_ride(a, c, lane);
  unsigned xf_tag = 0u;
  conv_u64_t xf_pre[NT][2];                            // the lane's granules, peeked at the start of the last chunk of the first item: landed by the epilogue
  bool xf_pending = false;                             // epi_mode 4 / 5 with xf_tab: (scale, shift) of this lane's channels are polled before the first epilogue
  int xf_nparts = 0;
  const bool xf_epi = (a.xf_tab != nullptr) && (a.epi_mode == 4 || a.epi_mode == 5);      // the coefficients feed this launch's EPILOGUE (residual tail)
  if (a.xf_tab != nullptr) {
    xfin_header(a, xf_tag, xf_nparts);
    xf_pending = xf_epi;
    if (!xf_epi) {
      // the coefficients feed this launch's PROLOGUE: they must be in LDS before the staging waves store their first chunk (barrier #0).  The staging waves
      // have issued that chunk's global loads already; the MFMA waves reduce their channels, then fill the table from the published granules.
      if (!xfin_produce(a, xf_tag, xf_nparts)) __builtin_amdgcn_s_sleep(30);               // (~1 us: a first poll before any publisher can be done is a wasted round trip)
      if (a.pro_mode == 2) xfin_fill<3>(a, cf_lds, a.cin_pad, xf_tag, vb & (kXfinRep - 1), MS_TID, 256, 1.f, 0.f);
      else xfin_fill<2>(a, cf_lds, a.cin_pad, xf_tag, vb & (kXfinRep - 1), MS_TID, 256, 1.f, 0.f);
    }
  }
  // (no s_setprio here: the STAGING waves get the priority - measured 290.7 -> 295.2 steps/s against the opposite choice; a staging wave that
  //  loses issue arbitration to back-to-back MFMAs is what the MFMA waves end up waiting for at the barrier)       // the MFMA-issuing wave wins issue arbitration against the staging wave of its SIMD
  const int m = lane & 15, k = lane >> 4;
  constexpr int MT = G::MT;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // M-tile i of this wave: NARROW: rows 2*wave+i, columns 0..15; else rows 2*wave+(i>>1), columns (i&1)*16..
  const int a_lane = k * PS + m + wave * (NARROW ? 1 : 2) * STRIDE * RS;     // all per-MFMA offsets below are immediates
  const int b_lane = k * WS + m;
  auto mt_row = [&](int i) { return NARROW ? (wave + i) : (wave * 2 + (i >> 1)); };
  auto mt_col = [&](int i) { return NARROW ? 0 : ((i & 1) * 16); };

  // FULL = every channel group of the chunk is live: straight-line code (no guards), so the LDS reads of later steps are
  // issued ahead of the MFMAs that consume earlier ones; the guarded form only runs for a layer's ragged last chunk.
  auto compute = [&](const float* buf, auto full_tag, int ncg) {
    constexpr bool FULL = decltype(full_tag)::value;
    const float* ap = buf + a_lane;
    const float* bp = buf + CK * PS + b_lane;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
      for (int cg = 0; cg < CK / 4; ++cg) {
        if (FULL || cg < ncg) {
          float bf[NT], af[MT];
#pragma unroll
          for (int j = 0; j < NT; ++j) bf[j] = bp[(tap * CK + cg * 4) * WS + j * 16];
#pragma unroll
          for (int i = 0; i < MT; ++i)
            af[i] = ap[cg * 4 * PS + G::tap_off(tap) + (NARROW ? i : (i >> 1)) * STRIDE * RS + (NARROW ? 0 : (i & 1) * 16)];
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  // ---- epilogue of one finished item (registers + global stores only) ----
  // bias of this lane's output channels, re-read only when the channel block changes (a global load inside the epilogue
  // would expose a full L2 round trip per item on the MFMA wave's critical path)
  // running per-wave BatchNorm statistics (count, mean, M2) of every pixel this wave produced, per output channel: a workgroup keeps
  // ONE channel block for its whole life (grid is a multiple of ncb), so the partial table gets one slot per wave, not one per tile
  float st_n = 0.f, st_mean[NT], st_m2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { st_mean[j] = 0.f; st_m2[j] = 0.f; }
  float bias_v[NT];
  float mk_sc[NT], mk_sh[NT], mk_mu[NT];              // epi_mode 3: forward BatchNorm map + mean of this lane's channels
  int bias_co0 = -1;
  auto load_bias = [&](int co0) {
    if (co0 == bias_co0) return;
    bias_co0 = co0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = co0 + j * 16 + m;
      int bidx = co;
      if (a.epi_mode == 2) bidx = co % a.cout_real;
      bias_v[j] = (a.bias != nullptr && co < a.Cout) ? a.bias[bidx] : 0.f;
      if (a.epi_mode >= 3 && !xf_pending) {
        const float4 cf = (co < a.Cout) ? reinterpret_cast<const float4*>(a.mk_coef)[co] : make_float4(0.f, 0.f, 0.f, 0.f);
        mk_sc[j] = cf.x; mk_sh[j] = cf.y; mk_mu[j] = cf.z;
      }
    }
  };
  auto epilogue = [&](int n, int tile, int co0) {
    if (xf_pending) {          // first epilogue of this workgroup (it keeps one channel block for its life): the coefficients some wave of the launch published
      xf_pending = false;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = co0 + j * 16 + m;
        const float2 cf = (co < a.Cout) ? xfin_poll(a, co, vb & (kXfinRep - 1), xf_tag, xf_pre[j]) : make_float2(0.f, 0.f);
        mk_sc[j] = cf.x; mk_sh[j] = cf.y; mk_mu[j] = 0.f;
      }
    }
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int xq = 4 * k;     // D layout (16x16): column (output channel) = lane&15, rows (pixels) = 4*(lane>>4) + reg
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] += bias_v[j];
    const bool full = (oy0 + TH <= a.Hout) && (ox0 + TW <= a.Wout);     // no masking needed (wave-uniform)
    if (a.stats != nullptr) {
      // PER-LANE running (count, mean, M2) of the pixels this lane produced (4 per M-tile), Chan-merged tile by tile; the four lanes that share a
      // channel are merged once at the end of the kernel (no ds_bpermute chain and no IEEE division per item - see ms_conv_wide.h)
      float cnt = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int y = oy0 + mt_row(i);
        const int nx = min(4, a.Wout - (ox0 + mt_col(i) + xq));
        if (full) cnt += 4.f;
        else if (y < a.Hout && nx > 0) cnt += (float)nx;
      }
      if (cnt > 0.f) {
        const float rc = __builtin_amdgcn_rcpf(cnt);
        const float nt_ = st_n + cnt;
        const float wgt = cnt * __builtin_amdgcn_rcpf(nt_);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int y = oy0 + mt_row(i);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int x = ox0 + mt_col(i) + xq + r;
              s += (full || ((y < a.Hout) && (x < a.Wout))) ? acc[i][j][r] : 0.f;
            }
          }
          const float mean = s * rc;
          float q = 0.f;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int y = oy0 + mt_row(i);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int x = ox0 + mt_col(i) + xq + r;
              const float d = acc[i][j][r] - mean;
              q += (full || ((y < a.Hout) && (x < a.Wout))) ? d * d : 0.f;
            }
          }
          const float d = mean - st_mean[j];
          st_mean[j] += d * wgt;
          st_m2[j] += q + d * d * st_n * wgt;
        }
        st_n = nt_;
      }
    }
    if (a.epi_mode == 3) {
      // g = acc * lrelu'(sc*u + sh); running sums of g and g*(u - mean) per channel in st_mean / st_m2 (act_bwd_reduce_kernel<1>)
      const bool vec4 = full && (a.Wout % 4 == 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = co0 + j * 16 + m;
        if (co >= a.Cout) continue;
        const size_t pb = ((size_t)n * a.Cout + co) * a.Hout * a.Wout;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int y = oy0 + mt_row(i);
          const int x = ox0 + mt_col(i) + xq;
          if (y >= a.Hout) continue;
          const size_t off = pb + (size_t)y * a.Wout + x;
          float uu[4];
          if (vec4) { const float4 t = IO::ld4(a.mk_u, off); uu[0] = t.x; uu[1] = t.y; uu[2] = t.z; uu[3] = t.w; }
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) uu[r] = (x + r < a.Wout) ? IO::ld1(a.mk_u, off + r) : 0.f;
          }
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[i][j][r] * ((mk_sc[j] * uu[r] + mk_sh[j] > 0.f) ? 1.f : a.mk_slope);
            if (!vec4 && x + r >= a.Wout) v[r] = 0.f;
          }
          if (vec4) IO::st4(a.out, off, make_float4(v[0], v[1], v[2], v[3]));
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (x + r < a.Wout) IO::st1(a.out, off + r, v[r]);
          }
          s1 += (v[0] + v[1]) + (v[2] + v[3]);
          s2 += (v[0] * (uu[0] - mk_mu[j]) + v[1] * (uu[1] - mk_mu[j])) + (v[2] * (uu[2] - mk_mu[j]) + v[3] * (uu[3] - mk_mu[j]));
        }
        st_mean[j] += s1; st_m2[j] += s2;
      }
    } else if (a.epi_mode == 4 || a.epi_mode == 5) {
      // Tail of a residual block in ONE launch (encoder_decoder.py:62-64, 344-346: last_act(conv_input(x) + conv(x))): this launch is the 1x1 skip
      // convolution; its epilogue reads the raw output u of the block's second 3x3 conv, applies that layer's BatchNorm (sc, sh) and the activation:
      //   out = lrelu((sc*u + sh) + (acc + bias))          - the arithmetic of ms_bn_act on a materialised skip tensor, bit for bit.
      // mode 5: the skip conv ran at HALF resolution (a 1x1 conv commutes with nearest up-sampling): every value feeds a 2x2 block of outputs.
      const bool up2 = (a.epi_mode == 5);
      const int Ho = up2 ? 2 * a.Hout : a.Hout, Wo = up2 ? 2 * a.Wout : a.Wout;
      const bool vec4 = (a.Wout % 4 == 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = co0 + j * 16 + m;
        if (co >= a.Cout) continue;
        const float sc = mk_sc[j], sh = mk_sh[j];
        const size_t pb = ((size_t)n * a.Cout + co) * Ho * Wo;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int y = oy0 + mt_row(i);
          const int x = ox0 + mt_col(i) + xq;
          if (y >= a.Hout || x >= a.Wout) continue;
          if (!up2) {
            const size_t off = pb + (size_t)y * Wo + x;
            if (vec4) {
              const float4 t = IO::ld4(a.mk_u, off);
              float4 o;
              o.x = leaky((sc * t.x + sh) + acc[i][j][0], a.mk_slope); o.y = leaky((sc * t.y + sh) + acc[i][j][1], a.mk_slope);
              o.z = leaky((sc * t.z + sh) + acc[i][j][2], a.mk_slope); o.w = leaky((sc * t.w + sh) + acc[i][j][3], a.mk_slope);
              IO::st4(a.out, off, o);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (x + r < a.Wout) IO::st1(a.out, off + r, leaky((sc * IO::ld1(a.mk_u, off + r) + sh) + acc[i][j][r], a.mk_slope));
            }
          } else {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
              const size_t off = pb + (size_t)(2 * y + dy) * Wo + 2 * x;
              if (vec4) {
                const float4 t0 = IO::ld4(a.mk_u, off), t1 = IO::ld4(a.mk_u, off + 4);
                float4 o0, o1;
                o0.x = leaky((sc * t0.x + sh) + acc[i][j][0], a.mk_slope); o0.y = leaky((sc * t0.y + sh) + acc[i][j][0], a.mk_slope);
                o0.z = leaky((sc * t0.z + sh) + acc[i][j][1], a.mk_slope); o0.w = leaky((sc * t0.w + sh) + acc[i][j][1], a.mk_slope);
                o1.x = leaky((sc * t1.x + sh) + acc[i][j][2], a.mk_slope); o1.y = leaky((sc * t1.y + sh) + acc[i][j][2], a.mk_slope);
                o1.z = leaky((sc * t1.z + sh) + acc[i][j][3], a.mk_slope); o1.w = leaky((sc * t1.w + sh) + acc[i][j][3], a.mk_slope);
                IO::st4(a.out, off, o0);
                IO::st4(a.out, off + 4, o1);
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  if (x + r < a.Wout) {
                    IO::st1(a.out, off + 2 * r, leaky((sc * IO::ld1(a.mk_u, off + 2 * r) + sh) + acc[i][j][r], a.mk_slope));
                    IO::st1(a.out, off + 2 * r + 1, leaky((sc * IO::ld1(a.mk_u, off + 2 * r + 1) + sh) + acc[i][j][r], a.mk_slope));
                  }
                }
              }
            }
          }
        }
      }
    } else if (a.epi_mode == 2 && NT >= 2 && (a.cout_real == 16 || (a.cout_real == 32 && NT == 4)) && (a.Wout % 4 == 0) &&
               ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0)) {
      // ConvTranspose2d k=2 s=2, paired form: the columns of dx = 0 and dx = 1 of one (dy, co) sit in the same lane, 16*C16 columns apart
      // (the tile starts at an even (dy,dx) block), so the 4 pixels of a lane become 8 CONSECUTIVE output floats: two 16-byte stores instead of
      // eight 4-byte stores at stride 8 (16->4x16 @16x128x128: the scattered form ran at 1.4 TB/s)
      const int Ho = 2 * a.Hout, Wo = 2 * a.Wout;
      const int C16 = a.cout_real / 16;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bool is_dx0 = (C16 == 1) ? ((j & 1) == 0) : ((j & 2) == 0);
        if (!is_dx0) continue;
        const int col = co0 + j * 16 + m;
        const int q = col / a.cout_real, co = col - q * a.cout_real;
        const int dy = q >> 1;
        const size_t op = ((size_t)n * a.cout_real + co) * Ho * Wo;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int y = oy0 + mt_row(i);
          const int x = ox0 + mt_col(i) + xq;
          if (y >= a.Hout || x >= a.Wout) continue;
          const size_t o = op + (size_t)(2 * y + dy) * Wo + 2 * x;
          // (C16 is wave-uniform: both selects compile to one of the two register sets)
          const f32x4 p0 = acc[i][j], p1 = (C16 == 1) ? acc[i][min(j + 1, NT - 1)] : acc[i][min(j + 2, NT - 1)];
          IO::st4(a.out, o, make_float4(p0[0], p1[0], p0[1], p1[1]));
          IO::st4(a.out, o + 4, make_float4(p0[2], p1[2], p0[3], p1[3]));
        }
      }
    } else if (a.epi_mode == 2) {
      // ConvTranspose2d k=2 s=2: GEMM column j = (dy*2+dx)*cout_real + co -> out[n,co,2y+dy,2x+dx]
      const int Ho = 2 * a.Hout, Wo = 2 * a.Wout;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = co0 + j * 16 + m;
        if (col >= 4 * a.cout_real) continue;
        const int q = col / a.cout_real, co = col - q * a.cout_real;
        const int dy = q >> 1, dx = q & 1;
        const size_t op = ((size_t)n * a.cout_real + co) * Ho * Wo;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int y = oy0 + mt_row(i);
          if (y >= a.Hout) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int x = ox0 + mt_col(i) + xq + r;
            if (x < a.Wout) IO::st1(a.out, op + (size_t)(2 * y + dy) * Wo + 2 * x + dx, acc[i][j][r]);
          }
        }
      }
    } else if (full && (a.Wout % 4 == 0)) {
      // fast path: whole tile inside the image, 16-B stores, no per-element predicates
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = co0 + j * 16 + m;
        if (co >= a.Cout) continue;
        const size_t op = (((size_t)n * a.Cout + co) * a.Hout + oy0) * a.Wout + ox0 + xq;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const size_t o = op + (size_t)mt_row(i) * a.Wout + mt_col(i);
          float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          if (a.epi_mode == 1) { const float4 p = IO::ld4(a.out, o); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
          IO::st4(a.out, o, v);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = co0 + j * 16 + m;
        if (co >= a.Cout) continue;
        const size_t op = ((size_t)n * a.Cout + co) * a.Hout * a.Wout;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int y = oy0 + mt_row(i);
          if (y >= a.Hout) continue;
          const int x = ox0 + mt_col(i) + xq;
          const size_t o = op + (size_t)y * a.Wout + x;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (x + r < a.Wout) IO::st1(a.out, o + r, (a.epi_mode == 1) ? (IO::ld1(a.out, o + r) + acc[i][j][r]) : acc[i][j][r]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  int item = vb, chunk = 0, n, tile, cb;
  decode(item, n, tile, cb);
  load_bias(cb * COUT_TILE);
  lds_barrier();                                      // barrier #0
  // the reducing waves of a cross-workgroup finalize work HERE, while the staging waves of their workgroup wait for the first chunk's global loads anyway:
  // in front of barrier #0 the ~4.5 us of cold table loads + publish delayed the whole workgroup's pipeline, and with a static work split the launch ends
  // with its slowest workgroup (measured: +3.9 us per launch, profiles/r03_experiments.txt)
  if (xf_epi) xfin_produce(a, xf_tag, xf_nparts);
  lds_barrier();                                      // barrier #1: chunk 0 is in buffer 0
  for (int p = 0; p < T; ++p) {
    const int c0 = chunk * CK;
    const int ncg = min(CK / 4, (a.cin_pad - c0) / 4);
    if (xf_pending && chunk + 1 == nchunks) {          // (wave-uniform) one round trip ahead of the first epilogue
#pragma unroll
      for (int j = 0; j < NT; ++j) xfin_peek(a, min(cb * COUT_TILE + j * 16 + m, a.xf_C - 1), vb & (kXfinRep - 1), xf_pre[j]);
    }
    if (!(a.dbg & 1)) {
      if (ncg == CK / 4) compute(smem + (p & 1) * BUF, std::true_type{}, CK / 4);
      else compute(smem + (p & 1) * BUF, std::false_type{}, ncg);
    }
    if (chunk + 1 == nchunks) {
      if (!(a.dbg & 4)) epilogue(n, tile, cb * COUT_TILE);
      chunk = 0; item += gridDim.x;
      if (p + 1 < T) { decode(item, n, tile, cb); load_bias(cb * COUT_TILE); }
    } else {
      ++chunk;
    }
    if (p + 1 < T) lds_barrier();                     // barrier #(p+2): chunk p+1 visible; this wave is done with chunk p
  }
  // table layout: [0] = {slots in use per channel}, then [1 + co*kStatSlots + slot]; one slot per workgroup of the channel block (conv_table_tail)
  if (a.stats != nullptr) conv_table_tail<NT, true>(a, smem, vb, ncb, st_n, st_mean, st_m2);
  else if (a.epi_mode == 3) conv_table_tail<NT, false>(a, smem, vb, ncb, 0.f, st_mean, st_m2);
}

// Resident 512-thread workgroups per CU of kernel `fn` with `lds_bytes` of dynamic LDS, as the occupancy API sees them (VGPRs AND AGPRs, SGPRs, LDS, wave slots), asked
// once per (instantiation, LDS size).  Every persistent conv grid is capped by it: the `_xfin` launches poll granules that other workgroups of the SAME launch publish,
// so the whole grid must be co-resident whatever order the dispatcher uses (hipFuncAttributes.numRegs alone does not count accumulation registers).
inline int conv_resident_per_cu(const void* fn, size_t lds_bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, size_t>, int> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find({fn, lds_bytes});
  if (it != cache.end()) return it->second;
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 512, lds_bytes) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
  cache[{fn, lds_bytes}] = n;
  return n;
}

template <int KS, int STRIDE, int FETCH, int NT, bool VEC, bool NARROW, bool IN2, typename AT>
int launch_conv_t(const ConvArgs& a, hipStream_t st) {
  using G = Geo<KS, STRIDE, FETCH, VEC, NARROW, NT>;
  const size_t lds_bytes = sizeof(float) * (2 * (size_t)G::BUF + 4 * (size_t)a.cin_pad);
  if (lds_bytes > 160 * 1024) { set_error("ms_conv2d: %d input channels exceed the LDS coefficient table", a.Cin); return MS_ERR_INVALID; }
  static std::once_flag attr_once;                     // one flag per instantiation (no unsynchronised mutable state in the ABI)
  static int reg_limit = 2;                            // written once, inside the call_once
  std::call_once(attr_once, []() {
    const void* fn = (const void*)conv_mfma_kernel<KS, STRIDE, FETCH, NT, VEC, NARROW, IN2, AT>;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.numRegs > 128) reg_limit = 1;
  });
  const long nitems = (long)a.N * a.tiles_x * a.tiles_y * a.ncb;
  // resident workgroups per CU: 512 threads = 2 waves per SIMD each -> at most 2 within 128 registers per wave, ONE above that (the 64-channel-tile variants:
  // a grid of two per CU ran as two rounds of a persistent kernel); LDS 160 KiB per CU.  The register count is a property of the instantiation: asked once.
  int per_cu = std::max(1, std::min(reg_limit, (int)((160 * 1024) / (lds_bytes + 256))));
  per_cu = std::max(1, std::min(per_cu, conv_resident_per_cu((const void*)conv_mfma_kernel<KS, STRIDE, FETCH, NT, VEC, NARROW, IN2, AT>, lds_bytes)));
  long nblocks = std::min<long>(nitems, (long)num_cus() * per_cu);
  if (nblocks > a.ncb) nblocks -= nblocks % a.ncb;      // every workgroup keeps one channel block: item % ncb == blockIdx % ncb
  dim3 grid((unsigned)nblocks), block(512);
  MS_LAUNCH((conv_mfma_kernel<KS, STRIDE, FETCH, NT, VEC, NARROW, IN2, AT>), grid, block, lds_bytes, st, a);
  return check_launch("conv_mfma");
}
template <int KS, int STRIDE, int FETCH, int NT, bool VEC, bool NARROW, bool IN2>
int launch_conv(const ConvArgs& a, hipStream_t st) {
  if (a.act_bf16) {
    // bf16 storage is built for the vector staging paths only (rows of W % 4 == 0 elements, 16-byte aligned tensors: every layer of the networks)
    if constexpr (VEC) return launch_conv_t<KS, STRIDE, FETCH, NT, VEC, NARROW, IN2, ms_bf16>(a, st);
    else { set_error("ms_conv2d_bf16: needs W %% 4 == 0 (W %% 2 for the fused-fetch variants) and 16-byte aligned tensors"); return MS_ERR_INVALID; }
  }
  return launch_conv_t<KS, STRIDE, FETCH, NT, VEC, NARROW, IN2, float>(a, st);
}

// dispatch entry points implemented in the ms_conv_inst*.hip translation units
int conv_dispatch_k3s1(const ConvArgs& a, int fetch, int nt, bool vec, bool narrow, bool in2, hipStream_t st);
int conv_dispatch_k1s1(const ConvArgs& a, int nt, bool vec, bool narrow, bool in2, hipStream_t st);
int conv_dispatch_s2(const ConvArgs& a, int ks, int nt, bool vec, bool narrow, hipStream_t st);

}  // namespace ms
