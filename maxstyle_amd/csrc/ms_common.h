// Shared device/host helpers for the MaxStyle gfx950 kernels.
// CDNA4 only: 64-lane wavefronts are hard-coded (no warpSize indirection, no other back-ends).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MS_OK 0
#define MS_ERR_INVALID (-1)      // bad argument / unsupported shape
#define MS_ERR_ALIGN (-2)        // pointer not aligned as the entry point requires
#define MS_ERR_WORKSPACE (-3)    // caller-provided workspace too small

// hipGetLastError() is per-thread state shared with every other HIP user in the process (PyTorch leaves benign errors behind,
// e.g. from its device-availability probe): clear it right before our launch so check_launch() reports OUR launch only.
#define MS_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace ms {

constexpr int kWave = 64;
constexpr int kNumCU = 256;      // MI355X: 8 XCDs x 32 CUs - CAPACITY of the per-workgroup tables; grids are sized from num_cus()
int num_cus();                   // compute units of the current device (hipDeviceProp, read once per device; <= kNumCU)
constexpr int kStatSlots = kNumCU * 2 * 4;   // per-channel capacity (row stride) of the conv kernels' partial tables; one slot per workgroup of the channel block is used

// last error string (thread-local; the ABI itself never throws)
void set_error(const char* fmt, ...);

// ---- library options (ms_set_option / ms_get_option, include/maxstyle_hip.h) ------------------------------------------------------------------------
// The ONLY process-wide knobs of the library: which kernel form the dispatch picks where more than one is built (every form computes the same convolution; the
// defaults are the product, the others are kept for A/B timing and for the "same bits" tests).  Nothing in the library reads the environment: a caller sets an
// option explicitly.  Relaxed atomics: a setter racing a launch gets one form or the other, never a torn state.
enum Opt {
  OPT_CONV_WIDE = 0,      // "conv.wide"        1: 3x3 stride-1 convs with rows >= 64 (Winograd forms: >= 20) pixels take conv_wide_kernel; 0: the first-generation kernel everywhere
  OPT_CONV_WINO,          // "conv.wino"        1: a caller's MS_FETCH_WINOGRAD opt-in is honoured; 0: ignored (direct form); 2: Winograd form forced wherever legal
  OPT_CONV_WINO32,        // "conv.wino32"      1: the 8 x 32-pixel Winograd tile for rows of 20..63 pixels
  OPT_CONV_WINO_NT,       // "conv.wino_nt"     0: automatic; 1 / 2: channel blocks per staged Winograd tile
  OPT_CONV_WINO_BLOCK,    // "conv.wino_block"  1: block form by fill; 0: never; 2: wherever legal
  OPT_CONV_WIDE_ROWS,     // "conv.wide_rows"   0: automatic; 4 / 8: rows per tile of the direct-form wide kernel
  OPT_CONV_K1S,           // "conv.k1s"         1: streaming 1x1 form where eligible
  OPT_CONV_K1G,           // "conv.k1g"         1: LDS-tiled GEMM 1x1 form where eligible
  OPT_CONV_S2G2,          // "conv.s2g2"        1: second-generation stride-2 3x3 form where eligible
  OPT_CONV_K3N,           // "conv.k3n"         1: second-generation 3x3 form for rows of 12 / 14 / 16 pixels where eligible (ms_conv_k3n.h)
  OPT_CONV_K9,            // "conv.k9"          ms_conv3x3_small_cin at Cin = 1: 1: the nine taps as the K dimension of the MFMA (3 per 16 pixels, ms_conv2d's bits); 0: the vector-ALU form
  OPT_CONV_FORCE_NT,      // "conv.force_nt"    0: automatic; 1 / 2 / 4: output-channel blocks of 16 per workgroup (tuning: tools/tune_conv.py)
  OPT_STYLE_FUSED,        // "style.fused"      1: single-read MaxStyle kernel where its grid fits the chip; 0: three-launch path
  OPT_CONV_WINO_FLAT,     // "conv.wino_flat"   1: the flattened-tile Winograd form on images of 20 / 24 / 28-pixel rows where it saves a round of the grid (ms_f32wf_t); 0: never; 2: wherever legal
  OPT_DIAG_CONV_DBG,      // "diag.conv_dbg"    timing-only ablation bits of the conv kernels (results are WRONG with any bit set): 1 no MFMA loop, 2 no global loads, 4 no stores, 8 no LDS stores, 16 no epilogue
  OPT_COUNT
};
int opt(int which);
long long* conv_trace_buffer();      // cycle-stamp destinations of the -DMS_CONV_TRACE_BUILD / -DMS_WGRAD_TRACE_BUILD diagnostic builds (ms_diag_set_trace); null otherwise
long long* wgrad_trace_buffer();

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return MS_OK;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (<= 16 waves). `red` is >= 16 floats of LDS.
// Every thread gets the total. Contains two barriers; safe to call repeatedly with the same `red`.
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();               // protect `red` from the previous call's readers
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];   // fixed order: deterministic
  return t;
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// Chan et al. pairwise merge of (count, mean, M2) in fp64.
__device__ __forceinline__ void chan_merge(double& n, double& mean, double& m2, double nb, double meanb, double m2b) {
  if (nb == 0.0) return;
  if (n == 0.0) { n = nb; mean = meanb; m2 = m2b; return; }
  const double nt = n + nb, d = meanb - mean;
  mean += d * (nb / nt);
  m2 += m2b + d * d * (n * nb / nt);
  n = nt;
}

// LeakyReLU / ReLU for 0 <= slope <= 1 (what every caller passes; the C entry points check it): max(v, v*slope) is the same value bit for bit as the select
// form (v > 0 ? v : v*slope; -0 -> -0, NaN -> NaN) in two vector instructions instead of three - it runs in the staging waves of the conv kernels, where
// every vector instruction waits behind an MFMA.
__device__ __forceinline__ float leaky(float v, float slope) { return fmaxf(v, v * slope); }

// ---------------------------------------------------------------------------------------------------------------------------------------
// Activation STORAGE type of the conv stack (BASELINE config 5: bf16 activations): kernels are templated on AT = float or ms_bf16 and touch
// activation tensors only through ActIO<AT> - element offsets in, fp32 values out.  Statistics, coefficients, accumulators, LDS tiles and
// the matrix arithmetic stay fp32 whatever AT is; a bf16 store rounds to nearest even (v_cvt_pk_bf16_f32).  For AT = float every helper is
// the plain load / store it replaces (same instructions).  Vector forms need the element offset to be a multiple of the vector length and the
// tensor base 16-byte aligned, as before.
struct ms_bf16 { uint16_t v; };
typedef __bf16 ms_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ms_pack_bf16x2(float a, float b) {
  ms_bf16x2_t p; p[0] = (__bf16)a; p[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ uint16_t ms_to_bf16(float a) { return __builtin_bit_cast(uint16_t, (__bf16)a); }
template <typename AT> struct ActIO;
template <> struct ActIO<float> {
  static constexpr int kBytes = 4;
  static __device__ __forceinline__ float ld1(const void* b, size_t o) { return reinterpret_cast<const float*>(b)[o]; }
  static __device__ __forceinline__ float2 ld2(const void* b, size_t o) { return *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(b) + o); }
  static __device__ __forceinline__ float4 ld4(const void* b, size_t o) { return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(b) + o); }
  static __device__ __forceinline__ void st1(void* b, size_t o, float v) { reinterpret_cast<float*>(b)[o] = v; }
  static __device__ __forceinline__ void st2(void* b, size_t o, float2 v) { *reinterpret_cast<float2*>(reinterpret_cast<float*>(b) + o) = v; }
  static __device__ __forceinline__ void st4(void* b, size_t o, float4 v) { *reinterpret_cast<float4*>(reinterpret_cast<float*>(b) + o) = v; }
};
template <> struct ActIO<ms_bf16> {
  static constexpr int kBytes = 2;
  static __device__ __forceinline__ float up(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
  static __device__ __forceinline__ float ld1(const void* b, size_t o) { return up(reinterpret_cast<const uint16_t*>(b)[o]); }
  static __device__ __forceinline__ float2 ld2(const void* b, size_t o) {
    const unsigned w = *reinterpret_cast<const unsigned*>(reinterpret_cast<const uint16_t*>(b) + o);
    return make_float2(__uint_as_float(w << 16), __uint_as_float(w & 0xFFFF0000u));
  }
  static __device__ __forceinline__ float4 ld4(const void* b, size_t o) {
    const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(b) + o);
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xFFFF0000u));
  }
  static __device__ __forceinline__ void st1(void* b, size_t o, float v) { reinterpret_cast<uint16_t*>(b)[o] = ms_to_bf16(v); }
  static __device__ __forceinline__ void st2(void* b, size_t o, float2 v) { *reinterpret_cast<unsigned*>(reinterpret_cast<uint16_t*>(b) + o) = ms_pack_bf16x2(v.x, v.y); }
  static __device__ __forceinline__ void st4(void* b, size_t o, float4 v) {
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(b) + o) = make_uint2(ms_pack_bf16x2(v.x, v.y), ms_pack_bf16x2(v.z, v.w));
  }
};

// ms_bf16m: bf16 storage AND bf16 matrix arithmetic (v_mfma_f32_16x16x16_bf16, fp32 accumulation) - the wide conv kernel's "bf16 MFMA" mode: the operands
// of the contraction (prologue outputs, weights) are rounded to bf16 on their way into LDS.  Loads / stores are those of ms_bf16.
struct ms_bf16m { uint16_t v; };
template <> struct ActIO<ms_bf16m> : ActIO<ms_bf16> {};

// ms_f32w: fp32 storage, fp32 matrix arithmetic on the WINOGRAD F(2x2, 3x3) form of the 3x3 stride-1 convolution (wide conv kernel, "Winograd mode"):
// 16 products per 2x2 output pixels instead of 36.  Loads / stores are those of float.
struct ms_f32w { float v; };
template <> struct ActIO<ms_f32w> : ActIO<float> {};
// ms_bf16w / ms_bf16w32: bf16 STORAGE (loads / stores of ms_bf16), fp32 Winograd arithmetic
struct ms_bf16w { uint16_t v; };
template <> struct ActIO<ms_bf16w> : ActIO<ms_bf16> {};
struct ms_bf16w32 { uint16_t v; };
template <> struct ActIO<ms_bf16w32> : ActIO<ms_bf16> {};
// ms_f32w32: the same on 8-row x 32-pixel tiles (rows of 20..63 pixels)
struct ms_f32w32 { float v; };
template <> struct ActIO<ms_f32w32> : ActIO<float> {};
// ms_f32wb / ms_bf16wb: the Winograd form on BLOCKS - a workgroup's four MFMA waves take four independent 8x8-pixel blocks (4x4 tiles of 2x2 outputs each) from a
// flattened (image, block row, block column) enumeration, each staged with its own halo: any W, H that are multiples of 8 run without tile-quantisation waste
// (rows of 80 / 40 / 20 pixels fill the 64- and 32-pixel tiles to 62 %)
struct ms_f32wb { float v; };
template <> struct ActIO<ms_f32wb> : ActIO<float> {};
struct ms_bf16wb { uint16_t v; };
template <> struct ActIO<ms_bf16wb> : ActIO<ms_bf16> {};

// ms_f32wf_t<W> (round 6): the Winograd form on FLATTENED TILES for images of W = 20 / 24 / 28 pixels per row (config 4's deepest levels: encoder_decoder.py:22-74, 650-653 at FCN_64 widths) - the
// 2x2-output tiles of the whole batch form one list (image, tile row, tile column); a work item = 64 consecutive tiles (16 per MFMA wave) and the band of image rows they touch
// (of one image, or of the end of one and the start of the next): every MFMA row is a real tile (the 8-row x 32-pixel tile fills 52 % of its rows at 20 x 20)
template <int W> struct ms_f32wf_t { float v; };      // W = pixels per image row: 20 (config 4's deepest levels), 24 / 28 (the deep levels of the reference's shipped 192 / 224-pixel workloads)
template <int W> struct ActIO<ms_f32wf_t<W>> : ActIO<float> {};
typedef ms_f32wf_t<20> ms_f32wf;
template <typename T> struct ms_wf_width { static constexpr int value = 0; };
template <int W> struct ms_wf_width<ms_f32wf_t<W>> { static constexpr int value = W; };

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }


// ---- LDS-DMA by hand ------------------------------------------------------------------------------------------------------------------------------------
// buffer_load_dwordx4 ... lds (1 KB per wave-instruction: lane l's 16 bytes land at M0 + 16 l, no vector register).  Issued as INLINE ASSEMBLY, not through
// __builtin_amdgcn_raw_ptr_buffer_load_lds: the compiler's wait-count pass treats every later LDS access of the wave as a possible reader of the builtin's
// destination and puts `s_waitcnt vmcnt(0)` in front of the first ds_read / ds_write behind it (found in round 4: the mask prefetch of the Winograd kernel was waited for
// at once - its ~1 us of HBM latency exposed once per work item - and in the staging waves the wait also covered the NEXT chunk's global loads: no prefetch at all).
// The assembly is opaque to that pass; the code that reads the landing zone waits with its own counted `s_waitcnt vmcnt(N)` (loads return in order).
// M0 is written inside the statement and declared clobbered (any compiler-generated M0 user - movrel, sendmsg, another DMA builtin - must not assume it survives).
// rsrc: {base lo, base hi (stride 0), num_records, flags}; lds_byte_addr: wave-uniform LDS byte address; voff: per-lane byte offset; soff: wave-uniform byte offset.
typedef int ms_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ms_i32x4 ms_dma_rsrc(const void* base) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  ms_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(p & 0xFFFFFFFFull));
  r.y = __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xFFFFull));
  r.z = 0x7FFFFFFF;
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ ms_i32x4 ms_dma_rsrc_n(const void* base, unsigned bytes) {      // ... with the exact size: every offset >= bytes reads as 0
  ms_i32x4 r = ms_dma_rsrc(base);
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  return r;
}
__device__ __forceinline__ void ms_lds_dma16(ms_i32x4 rsrc, unsigned lds_byte_addr, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
// the 4-byte form: lane i's dword lands at lds_byte_addr + 4 i (256 contiguous bytes per wave instruction)
__device__ __forceinline__ void ms_lds_dma4(ms_i32x4 rsrc, unsigned lds_byte_addr, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_byte_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ unsigned ms_lds_addr(const void* p) {      // byte address inside the workgroup's LDS allocation of a pointer into a __shared__ array
  typedef __attribute__((address_space(3))) void* lds_p;
  return (unsigned)(unsigned long long)(lds_p)(const_cast<void*>(p));      // generic -> LDS address space (the low 32 bits of a flat LDS address are its offset)
}
}  // namespace ms
