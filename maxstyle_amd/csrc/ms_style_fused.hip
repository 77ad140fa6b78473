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
// Round-2 rewrite of the data path (the protocol is unchanged): the first version addressed its 16 float4 loads per thread with
// 64-bit flat pointers and per-load bounds guards; at 1024 threads (128 VGPRs) hipcc spilled 16 of the 64 data registers to
// scratch - every launch wrote and re-read 25 % of the tensor a second time (rocprofv3: WRITE_SIZE 83.9 MB against 67.1 MB
// algorithmic, `-Rpass-analysis=kernel-resource-usage`: "VGPRs Spill: 16, ScratchSize 68").  Now a unit is ONE buffer resource
// (base = start of the chunk, num_records = its bytes): every load / store is `buffer_load/store_dwordx4 v, voff, srsrc, soff`
// with one per-lane offset register and a scalar offset per register slot, out-of-range lanes are dropped by the hardware range
// check (no guards, no 64-bit address arithmetic), chunks are balanced (HW split evenly over S), and the kernel is built for
// 1024-, 512- and 256-thread workgroups so that small tensors still fill 256 CUs and two or four workgroups per CU overlap
// one's loads with another's stores.
//
// Progress: the grid never exceeds the workgroups that fit the chip together (device CU count x workgroups per CU at <= 128
// VGPRs), and a workgroup only ever waits for units of its own channel, which are the current or an earlier unit of other
// workgroups of the same launch (channel-major numbering + increasing processing order): the lowest unfinished channel always
// has every unit either done or being loaded, so it completes and releases its waiters.  That argument needs the launch to get
// the CUs it was sized for; a foreign kernel on another stream only delays it (its workgroups finish and free their slots).
// Every spin is bounded: a time-out sets the error word, which the host reads (ms_style_fused_status) - never silent zeros.
// Visibility: 8-byte agent-scope atomics on both sides (MI355X_MICROARCH.md "Valid forms", R2: the granule carries its own tag).
// Tags are LAUNCH EPOCHS: the state block holds an epoch word; every workgroup reads it at start (T = epoch + 1), publishes and accepts
// only granules tagged T, and the last workgroup to finish stores epoch = T.  Nothing is re-initialised between launches - an earlier
// version cleared the table with hipMemsetAsync before every launch, and under HIP-graph replay the kernel was observed polling
// granules the memset node had not cleared yet (stale or foreign words with a non-zero tag -> wrong statistics, NaN).  The state block
// is zero-filled ONCE by the caller and must stay dedicated to one layer.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "ms_common.h"
#include "maxstyle_hip.h"

namespace ms {

constexpr unsigned kSpinLimit = 1u << 22;
constexpr int kFusedNV = 16;                      // float4 register slots per thread (64 floats)

typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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

// Activation storage type of x / y: fp32, or bf16 (`*_bf16` entry points: half the HBM bytes; every statistic, coefficient and the arithmetic stay fp32).
// One 16-byte buffer load / store per register slot either way: 4 fp32 or 8 bf16 values; a thread holds 64 values in fp32 registers.
typedef __bf16 ms_bf16x2 __attribute__((ext_vector_type(2)));
// A 16-byte buffer store with a REGISTER soffset must not have its data registers overwritten by the very next vector instruction: measured on MI355X in
// the wide conv epilogue (ms_conv_wide.h, bstore4), lanes 12-15 of every 16-lane row of the second data dword then carry the NEW value, and hipcc pads this
// hazard only for stores WITHOUT a register soffset.  The guard keeps the data registers live (and unmodified) across one idle issue slot behind the store.
static __device__ __forceinline__ void store_guard(const u32x4& o) { asm volatile("s_nop 0" :: "v"(o)); }
template <typename T> struct StyleIo;
template <> struct StyleIo<float> {
  static constexpr int EPL = 4;
  template <int AUX> static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int voff, int soff, float (&d)[4]) {
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    d[0] = __uint_as_float(w.x); d[1] = __uint_as_float(w.y); d[2] = __uint_as_float(w.z); d[3] = __uint_as_float(w.w);
  }
  template <int AUX> static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, int voff, int soff, const float (&d)[4]) {
    u32x4 o; o.x = __float_as_uint(d[0]); o.y = __float_as_uint(d[1]); o.z = __float_as_uint(d[2]); o.w = __float_as_uint(d[3]);
    __builtin_amdgcn_raw_buffer_store_b128(o, r, voff, soff, AUX);
    store_guard(o);
  }
};
struct ms_bf16_tag {};
template <> struct StyleIo<ms_bf16_tag> {
  static constexpr int EPL = 8;
  static __device__ __forceinline__ unsigned pack2(float a, float b) {       // round-to-nearest-even (v_cvt_pk_bf16_f32), NaN stays NaN
    ms_bf16x2 p; p[0] = (__bf16)a; p[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, p);
  }
  template <int AUX> static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int voff, int soff, float (&d)[8]) {
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    const unsigned q[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { d[2 * i] = __uint_as_float(q[i] << 16); d[2 * i + 1] = __uint_as_float(q[i] & 0xFFFF0000u); }
  }
  template <int AUX> static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, int voff, int soff, const float (&d)[8]) {
    u32x4 o; o.x = pack2(d[0], d[1]); o.y = pack2(d[2], d[3]); o.z = pack2(d[4], d[5]); o.w = pack2(d[6], d[7]);
    __builtin_amdgcn_raw_buffer_store_b128(o, r, voff, soff, AUX);
    store_guard(o);
  }
};

struct FusedArgs {
  const void* x; void* y; float* mu; float* sig; float* gamma_std; float* beta_std;
  const float* lmda; const float* gamma_noise; const float* beta_noise; const int64_t* perm;
  float* coefA; float* coefS; u64* part; int* arrive; int* counter; int* err;
  int compute_std, B, C, HW, S, chunk, nv;
  float eps;
};

// THREADS: workgroup size; AUXL / AUXS: cache-policy bits of the x loads / y stores (0 default, 2 = nt); T: storage type of x / y (float | ms_bf16_tag)
template <int THREADS, int AUXL, int AUXS, typename T>
__global__ __launch_bounds__(THREADS, 4) void style_fused_kernel(const FusedArgs a) {
  using IO = StyleIo<T>;
  constexpr int EPL = IO::EPL, NSLOT = 4 * kFusedNV / EPL, ESZ = 16 / EPL;      // values per 16-byte access, register slots per thread, bytes per value
  __shared__ float red[16];
  __shared__ double redd[16];
  __shared__ float smu[256], ssig[256];
  __shared__ float pmean[1024], pm2[1024];          // polled partials of the channel: B*S <= 1024 units (host-checked)
  __shared__ unsigned s_tag;
  const int tid = threadIdx.x;
  const int B = a.B, C = a.C, HW = a.HW, S = a.S, chunk = a.chunk, nv = a.nv;
  const int G = B * S, total = C * G;
  // `counter` is the epoch word, `arrive` counts finished workgroups of this launch
  if (tid == 0) {
    unsigned t = (unsigned)__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    s_tag = (t == 0u) ? 1u : t;                     // 0 is the zero-filled (never published) state
  }
  __syncthreads();
  const unsigned tag = s_tag;
  const int voff = tid * 16;                        // per-lane byte offset inside a register slot's THREADS*16-byte stripe
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int c = t / G, r = t - c * G, b = r / S, s = r - b * S;
    const int p = b * C + c;
    const int beg = s * chunk, cnt = min(HW - beg, chunk);                 // values; both multiples of EPL
    const size_t base = (size_t)p * HW + beg;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.x)) + base * ESZ, 0, cnt * ESZ, 0x00020000);
    float v[NSLOT][EPL];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) v[j][e] = 0.f;
      if (j < nv) IO::template load<AUXL>(rx, voff, j * THREADS * 16, v[j]);          // wave-uniform; lanes past the chunk read 0 (range check)
    }
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
#pragma unroll
      for (int e = 0; e < EPL; e += 4) sum += (v[j][e] + v[j][e + 1]) + (v[j][e + 2] + v[j][e + 3]);
    }
    const float n = (float)cnt;
    const float mean_c = block_sum(sum, red) / n;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      if ((j * THREADS + tid) * EPL < cnt) {
#pragma unroll
        for (int e = 0; e < EPL; e += 4) {
          const float d0 = v[j][e] - mean_c, d1 = v[j][e + 1] - mean_c, d2 = v[j][e + 2] - mean_c, d3 = v[j][e + 3] - mean_c;
          m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      }
    }
    m2 = block_sum(m2, red);
    u64* gran = a.part + 2 * ((size_t)c * G);        // granules of channel c: unit (bb, ss) -> index 2*(bb*S+ss) (+1)
    if (tid == 0) { publish_granule(gran + 2 * (b * S + s), mean_c, tag); publish_granule(gran + 2 * (b * S + s) + 1, m2, tag); }
    // one thread per granule polls until it is published (our own two come back from memory as well)
    for (int q = tid; q < 2 * G; q += THREADS) {
      float val;
      poll_granule(gran + q, val, tag, a.err);
      if (q & 1) pm2[q >> 1] = val; else pmean[q >> 1] = val;
    }
    __syncthreads();
    // every plane of channel c: Chan-merge its S chunk partials (chunk sizes follow from the geometry)
    for (int bb = tid; bb < B; bb += THREADS) {
      double nn = 0.0, mean = 0.0, mm2 = 0.0;
      for (int ss = 0; ss < S; ++ss) {
        const float pm = pmean[bb * S + ss], pq = pm2[bb * S + ss];
        const double cn = (double)(min(HW, (ss + 1) * chunk) - ss * chunk);
        chan_merge(nn, mean, mm2, cn, (double)pm, (double)pq);
      }
      smu[bb] = (float)mean;
      ssig[bb] = sqrtf((float)(mm2 / (double)(HW - 1)) + a.eps);
    }
    __syncthreads();
    float gs, bs;
    if (a.compute_std & 1) {
      double am = 0.0, as = 0.0;
      for (int bb = tid; bb < B; bb += THREADS) { am += (double)smu[bb]; as += (double)ssig[bb]; }
      const double mean_mu = block_sum_d(am, redd) / B;
      const double mean_sg = block_sum_d(as, redd) / B;
      double qm = 0.0, qs = 0.0;
      for (int bb = tid; bb < B; bb += THREADS) {
        const double d1 = (double)smu[bb] - mean_mu, d2 = (double)ssig[bb] - mean_sg;
        qm += d1 * d1; qs += d2 * d2;
      }
      qm = block_sum_d(qm, redd); qs = block_sum_d(qs, redd);
      bs = (float)sqrt(qm / (double)(B - 1));
      gs = (float)sqrt(qs / (double)(B - 1));
    } else {
      gs = a.gamma_std[c]; bs = a.beta_std[c];
    }
    const float m = smu[b], sg = ssig[b];
    float A = sg, Sh = m;
    if (a.lmda != nullptr) {
      const float lam = (a.compute_std & 2) ? a.lmda[b] : fminf(fmaxf(a.lmda[b], 0.f), 1.f);     // bit 1: MixStyle (no clamp)
      const int pb = (int)a.perm[b];
      A = sg * (1.f - lam) + ssig[pb] * lam;
      Sh = m * (1.f - lam) + smu[pb] * lam;
    }
    if (a.gamma_noise != nullptr) {
      A += a.gamma_noise[p] * gs;
      Sh += a.beta_noise[p] * bs;
    }
    if (tid == 0 && s == 0) {
      a.mu[p] = m; a.sig[p] = sg; a.coefA[p] = A; a.coefS[p] = Sh;
      if (b == 0 && (a.compute_std & 1)) { a.gamma_std[c] = gs; a.beta_std[c] = bs; }
    }
    if (a.y != nullptr) {       // y == NULL: statistics and coefficients only (the consumer applies y = A/sig * (x - mu) + S itself: ms_head_fwd_styled)
      const float sc = A / sg;
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.y) + base * ESZ, 0, cnt * ESZ, 0x00020000);
#pragma unroll
      for (int j = 0; j < NSLOT; ++j) {
        if (j < nv) {
          float o[EPL];
#pragma unroll
          for (int e = 0; e < EPL; ++e) o[e] = sc * (v[j][e] - m) + Sh;
          IO::template store<AUXS>(ry, voff, j * THREADS * 16, o);                       // lanes past the chunk are dropped by the range check
        }
      }
    }
    __syncthreads();      // smu/ssig are reused by the next unit
  }
  // end of launch: the last workgroup to get here advances the epoch (every workgroup has read it by then) and re-arms the counter
  if (tid == 0) {
    const int prev = __hip_atomic_fetch_add(a.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (int)gridDim.x - 1) {
      __hip_atomic_store(a.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.counter, (int)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct FusedPlan { bool ok; int threads, nv, chunk, S, grid; size_t part_off, bytes; };

// State block: ints [0] epoch, [1] error word, [2] finished-workgroup counter; granules (8 B each) from byte 16.
// Geometry: a unit is a balanced chunk of one plane held in registers by one workgroup (<= 64 floats per thread).  Prefer the fattest
// workgroup that still gives every CU work; all units of one channel (G = B*S) must be resident together.
static FusedPlan fused_plan(int B, int C, int HW, int epl = 4) {
  FusedPlan pl{};
  pl.ok = false;
  if (HW % epl != 0 || B < 2 || B > 256) return pl;
  const int cus = num_cus();
  constexpr int force_threads = 0, force_split = 0;      // (were A/B environment switches until round 5; the measured choice is the rule below)
  // candidates in order of preference.  Measured on MI355X at 16x16x256x256 (kernel-trace medians): 512-thread workgroups, two per CU, 25.8 us
  // (one's stores overlap the other's loads) against 27.0 us for one 1024-thread workgroup per CU and 30 us for four of 256; so: 512 threads
  // when the planes already give every slot a unit, else 1024 threads (fewer, fatter units), else split planes into chunks.
  struct Cand { int threads; bool split; };
  const Cand cand[4] = {{512, false}, {1024, true}, {512, true}, {256, true}};
  for (int k = 0; k < 4; ++k) {
    const int threads = cand[k].threads;
    if (force_threads && threads != force_threads) continue;
    const int per_cu = 1024 / threads;                                            // workgroups per CU at <= 128 VGPRs (4 waves per SIMD)
    const long resident = (long)cus * per_cu;
    const int max_chunk = threads * 4 * kFusedNV;                                  // 64 values per thread whatever the storage type
    int S = cdiv(HW, max_chunk);
    // split planes further while the tensor has fewer units than the chip has workgroup slots (small C: layer 5 has 16 planes)
    if (cand[k].split || force_threads)
      while ((long)B * C * S < resident && (long)B * (S * 2) <= resident && HW / (S * 2) >= threads * epl && B * S * 2 <= 1024) S *= 2;
    if (force_split) S = std::max(S, force_split);
    int chunk = (cdiv(HW, S) + epl - 1) / epl * epl;
    S = cdiv(HW, chunk);
    const int nv = cdiv(chunk, threads * epl);                                    // 16-byte register slots per thread
    const long G = (long)B * S;
    if (nv > 4 * kFusedNV / epl || G > resident || G > 1024) continue;
    const bool last = (k == 3) || force_threads;
    if (!last && (long)B * C * S < resident) continue;                            // the next candidate fills more CUs
    pl.threads = threads; pl.nv = nv; pl.chunk = chunk; pl.S = S;
    pl.grid = (int)std::min<long>((long)C * G, resident);                         // never more than fit the chip together
    pl.part_off = 16;                                                             // [0] epoch, [1] error word, [2] arrivals
    pl.bytes = pl.part_off + 2 * (size_t)C * G * sizeof(u64);                     // two tagged granules per unit
    pl.ok = true;
    return pl;
  }
  return pl;
}

// The state block must not depend on the A/B switches above in a way that moves it between calls: its size is the maximum over the
// candidate geometries (granules are indexed by the geometry of the launch; a launch only reads granules carrying its own epoch).
static size_t fused_state_bytes(int B, int C, int HW) {
  if (HW % 4 != 0 || B < 2 || B > 256) return 0;
  return 16 + 2 * (size_t)C * 1024 * sizeof(u64);
}

}  // namespace ms

using namespace ms;

extern "C" size_t ms_style_fused_ws_bytes(int B, int C, int HW) {
  const FusedPlan pl = fused_plan(B, C, HW);
  return pl.ok ? std::max(pl.bytes, fused_state_bytes(B, C, HW)) : 0;
}

extern "C" int ms_style_fused_plan(int B, int C, int HW, int* threads, int* nv, int* S, int* grid) {
  const FusedPlan pl = fused_plan(B, C, HW);
  if (!pl.ok) return MS_ERR_INVALID;
  if (threads) *threads = pl.threads;
  if (nv) *nv = pl.nv;
  if (S) *S = pl.S;
  if (grid) *grid = pl.grid;
  return MS_OK;
}

template <typename T>
static int style_fwd_fused_impl(const void* x, void* y, float* mu, float* sig, float* gamma_std, float* beta_std, int compute_std,
                                const float* lmda, const float* gamma_noise, const float* beta_noise, const int64_t* perm,
                                float* coefA, float* coefS, int B, int C, int HW, float eps, void* ws, size_t ws_bytes, void* stream) {
  const FusedPlan pl = fused_plan(B, C, HW, StyleIo<T>::EPL);
  if (!pl.ok) { set_error("ms_style_fwd_fused: shape B=%d C=%d HW=%d is not eligible (use ms_style_fwd)", B, C, HW); return MS_ERR_INVALID; }
  if (!aligned16(x) || !aligned16(y)) { set_error("ms_style_fwd_fused: x and y must be 16-byte aligned"); return MS_ERR_ALIGN; }
  if (ws == nullptr || ws_bytes < pl.bytes || !aligned16(ws)) { set_error("ms_style_fwd_fused: workspace too small or unaligned"); return MS_ERR_WORKSPACE; }
  if (lmda != nullptr && perm == nullptr) { set_error("ms_style_fwd_fused: mixing needs perm"); return MS_ERR_INVALID; }
  if ((gamma_noise == nullptr) != (beta_noise == nullptr)) { set_error("ms_style_fwd_fused: gamma/beta noise must both be given"); return MS_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  int* hdr = (int*)ws;
  FusedArgs a;
  a.x = x; a.y = y; a.mu = mu; a.sig = sig; a.gamma_std = gamma_std; a.beta_std = beta_std;
  a.lmda = lmda; a.gamma_noise = gamma_noise; a.beta_noise = beta_noise; a.perm = perm; a.coefA = coefA; a.coefS = coefS;
  a.part = (u64*)((char*)ws + pl.part_off); a.arrive = hdr + 2; a.counter = hdr; a.err = hdr + 1;
  a.compute_std = compute_std; a.B = B; a.C = C; a.HW = HW; a.S = pl.S; a.chunk = pl.chunk; a.nv = pl.nv; a.eps = eps;
  // cache policy (fp32): bit 0 nt loads of x, bit 1 nt stores of y.  A tensor that does not fit the 256 MB Infinity Cache cannot be found there by the next kernel
  // anyway: streaming it past the caches measured 184.8 -> 170.3 us at 16x64x320x320 (4.5 -> 4.9 TB/s, profiles/r02_experiments.txt section 4), while for the
  // cache-sized tensors of config 2 the default policy wins in the step (the next kernel reads y from the cache).
  const int policy = ((size_t)B * C * HW * sizeof(float) > ((size_t)256 << 20)) ? 3 : 0;
  dim3 grid(pl.grid), block(pl.threads);
#define MS_FUSED(TT, AL, AS) MS_LAUNCH((style_fused_kernel<TT, AL, AS, T>), grid, block, 0, st, a)
#define MS_FUSED_T(TT) if (std::is_same<T, float>::value) { switch (policy & 3) { case 1: MS_FUSED(TT, 2, 0); break; case 2: MS_FUSED(TT, 0, 2); break; \
                                                                                 case 3: MS_FUSED(TT, 2, 2); break; default: MS_FUSED(TT, 0, 0); } } else { MS_FUSED(TT, 0, 0); }
  if (pl.threads == 1024) { MS_FUSED_T(1024) } else if (pl.threads == 512) { MS_FUSED_T(512) } else { MS_FUSED_T(256) }
#undef MS_FUSED_T
#undef MS_FUSED
  return check_launch("style_fused");
}

extern "C" int ms_style_fwd_fused(const float* x, float* y, float* mu, float* sig, float* gamma_std, float* beta_std, int compute_std,
                                  const float* lmda, const float* gamma_noise, const float* beta_noise, const int64_t* perm,
                                  float* coefA, float* coefS, int B, int C, int HW, float eps, void* ws, size_t ws_bytes, void* stream) {
  return style_fwd_fused_impl<float>(x, y, mu, sig, gamma_std, beta_std, compute_std, lmda, gamma_noise, beta_noise, perm, coefA, coefS, B, C, HW, eps, ws, ws_bytes, stream);
}

// bf16 activations in, bf16 activations out (uint16_t = the bf16 bit pattern); statistics, coefficients and arithmetic fp32.  Needs H*W % 8 == 0.
extern "C" int ms_style_fwd_fused_bf16(const uint16_t* x, uint16_t* y, float* mu, float* sig, float* gamma_std, float* beta_std, int compute_std,
                                       const float* lmda, const float* gamma_noise, const float* beta_noise, const int64_t* perm,
                                       float* coefA, float* coefS, int B, int C, int HW, float eps, void* ws, size_t ws_bytes, void* stream) {
  return style_fwd_fused_impl<ms_bf16_tag>(x, y, mu, sig, gamma_std, beta_std, compute_std, lmda, gamma_noise, beta_noise, perm, coefA, coefS, B, C, HW, eps, ws, ws_bytes, stream);
}

extern "C" size_t ms_style_fused_ws_bytes_bf16(int B, int C, int HW) {
  const FusedPlan pl = fused_plan(B, C, HW, 8);
  return pl.ok ? std::max(pl.bytes, fused_state_bytes(B, C, HW)) : 0;
}

// Error word of the single-read kernel's state block (`ws` as handed to ms_style_fwd_fused): a bounded spin that timed out sets it and the
// launch's statistics are then invalid.  Device pointer: copy it with the caller's own (asynchronous) D2H, or pass host == 1 for a synchronous read
// (stream-ordered hipMemcpyAsync + hipStreamSynchronize) that also clears the word.
extern "C" int ms_style_fused_status(void* ws, int* out, void* stream) {
  if (ws == nullptr || out == nullptr) { set_error("ms_style_fused_status: null argument"); return MS_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  int* hdr = (int*)ws;
  hipError_t e = hipMemcpyAsync(out, hdr + 1, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { set_error("ms_style_fused_status: %s", hipGetErrorString(e)); return (int)e; }
  if (*out != 0) {
    e = hipMemsetAsync(hdr + 1, 0, sizeof(int), st);
    if (e != hipSuccess) { set_error("ms_style_fused_status: %s", hipGetErrorString(e)); return (int)e; }
    set_error("single-read MaxStyle kernel: a bounded spin timed out (the launch did not get the CUs it was sized for); its output is invalid");
  }
  return MS_OK;
}
