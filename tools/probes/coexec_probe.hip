// Does the fp32-input MFMA (v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD = the fp32 VECTOR rate) co-execute with VALU work on the same SIMD, or do they
// share the FMA lanes?  Build: hipcc -O3 --offload-arch=gfx950 tools/probes/coexec_probe.hip -o tools/probes/coexec_probe ; run on the GPU box.
// Modes (512-thread workgroups, 2 per CU unless stated):
//   0: every wave: N MFMAs                                   (baseline matrix time)
//   1: every wave: N x (1 MFMA + V independent v_fma_f32)    (same wave interleaved)
//   2: waves 0-3: N MFMAs, waves 4-7: N*V v_fma_f32          (partner waves on the same SIMDs)
//   3: every wave: N*V v_fma_f32 only                        (baseline vector time)
//   4: like 0 with the bf16 MFMA v_mfma_f32_16x16x16_bf16 ; 5: like 2 with the bf16 MFMA
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int V>
__device__ __forceinline__ void valu_block(float (&f)[8], float a, float b) {
#pragma unroll
  for (int v = 0; v < V; ++v) f[v & 7] = __builtin_fmaf(f[v & 7], a, b);
}

template <int MODE, int V>
__global__ __launch_bounds__(512) void probe(int iters, float* sink) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = 1.0f + i * 1e-3f;
  const float a = 1.0f + 1e-6f * (threadIdx.x & 15), b = 1e-7f * (threadIdx.x >> 4);
  bf16x8 ba, bb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(1.0f + i); bb[i] = (__bf16)(0.5f + i); }
  const bool mfma_wave = (MODE == 0 || MODE == 1 || MODE == 4) || ((MODE == 2 || MODE == 5) && wave < 4);
  const bool valu_wave = (MODE == 1 || MODE == 3) || ((MODE == 2 || MODE == 5) && wave >= 4);
  // the role of a wave is decided ONCE, outside the loops (a per-slot wave-role branch would dominate the timing)
  if (mfma_wave && valu_wave) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
        valu_block<V>(f, a, b);
      }
    }
  } else if (mfma_wave) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (MODE == 4 || MODE == 5) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[j & 3], 0, 0, 0);
        else acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
      }
    }
  } else if (valu_wave) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) valu_block<V>(f, a, b);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i];
  if (s == 123.456f) sink[0] = s;
}

template <int MODE, int V>
static float run(int iters, int wgs, float* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, V><<<wgs, 512>>>(iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE, V><<<wgs, 512>>>(iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* sink; hipMalloc(&sink, 4);
  const int iters = 2000, wgs = 512;
  const double n_mfma = (double)wgs * 8 * iters * 16;
  printf("512 WGs x 512 threads, %d x 16 MFMA slots per wave\n", iters);
  float t0 = run<0, 0>(iters, wgs, sink);
  printf("mode0 f32 MFMA only (8 waves/WG):        %8.1f us  -> %.1f TFLOP/s\n", t0, n_mfma * 2048 / t0 / 1e6);
  float t3 = run<3, 4>(iters, wgs, sink);
  printf("mode3 VALU only, 4 fma per slot:         %8.1f us\n", t3);
  float t1 = run<1, 4>(iters, wgs, sink);
  printf("mode1 same wave: MFMA + 4 fma per slot:  %8.1f us   (sum %.1f, max %.1f)\n", t1, t0 + t3, t0 > t3 ? t0 : t3);
  float t1b = run<1, 2>(iters, wgs, sink);
  float t3b = run<3, 2>(iters, wgs, sink);
  printf("mode1 same wave: MFMA + 2 fma per slot:  %8.1f us   (VALU alone %.1f)\n", t1b, t3b);
  float t2 = run<2, 4>(iters, wgs, sink);
  printf("mode2 waves0-3 MFMA, waves4-7 4 fma/slot:%8.1f us   (MFMA half alone %.1f, VALU half alone %.1f)\n", t2, t0 / 2, t3 / 2);
  float t2b = run<2, 8>(iters, wgs, sink);
  printf("mode2 waves0-3 MFMA, waves4-7 8 fma/slot:%8.1f us\n", t2b);
  float t4 = run<4, 0>(iters, wgs, sink);
  printf("mode4 bf16 MFMA 16x16x32 only:           %8.1f us  -> %.1f TFLOP/s\n", t4, n_mfma * 16384 / t4 / 1e6);
  float t5 = run<5, 4>(iters, wgs, sink);
  printf("mode5 waves0-3 bf16 MFMA, waves4-7 4 fma:%8.1f us   (MFMA half alone %.1f, VALU half alone %.1f)\n", t5, t4 / 2, t3 / 2);
  return 0;
}
