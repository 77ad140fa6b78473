// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13]; gfx9 layout)
//   hipcc --offload-arch=gfx950 -O2 -o simd_probe simd_probe.hip && ./simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
  extern __shared__ float smem[];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    out[blockIdx.x * 8 + wave] = hw;
  }
  // stay resident for a while so that two workgroups share a CU as in the conv kernels
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
  smem[threadIdx.x] = 0.f;
}
int main() {
  const int grid = 512;
  unsigned* d; hipMalloc(&d, grid * 8 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 72 * 1024, 0, d);      // 72 KB of LDS: two workgroups per CU
  hipDeviceSynchronize();
  unsigned h[grid * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int hist[8][4] = {};
  int pattern_rr = 0, pattern_other = 0;
  for (int b = 0; b < grid; ++b) {
    bool rr = true;
    const int s0 = (h[b * 8] >> 4) & 3;
    for (int w = 0; w < 8; ++w) { const int s = (h[b * 8 + w] >> 4) & 3; hist[w][s]++; if (s != ((s0 + w) & 3)) rr = false; }
    if (rr) ++pattern_rr; else ++pattern_other;
  }
  printf("workgroups with round-robin wave->SIMD placement: %d of %d (other: %d)\n", pattern_rr, grid, pattern_other);
  for (int w = 0; w < 8; ++w) printf("wave %d: SIMD histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  for (int b = 0; b < 4; ++b) { printf("wg %d:", b); for (int w = 0; w < 8; ++w) printf(" simd%u/cu%u", (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15); printf("\n"); }
  return 0;
}
