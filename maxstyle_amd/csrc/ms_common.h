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

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace ms
