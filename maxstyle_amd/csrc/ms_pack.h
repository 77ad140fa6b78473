// Device function shared by ms_wino_pack (ms_conv.hip) and the batched appendix refresh ms_appendix_batch (ms_conv_subpix.hip).
#pragma once
#include "ms_common.h"

namespace ms {
// one thread per (input channel, output channel) pair: U = G g G^T with EXACTLY the expression of the staging waves (conv_wide_kernel store_chunk), so the kernel that
// copies U from the appendix computes the same bits as the one that transforms the taps itself
__device__ __forceinline__ void wino_pack_one(long id, float* __restrict__ wp, int Cin, int Cout, int cin_pad, int cout_pad) {
  const int nchunks = Cin / 8, ncb = (Cout + 15) / 16;
  const long total = (long)ncb * nchunks * 128;
  if (id >= total) return;
  const int m = (int)(id & 15), ci = (int)((id >> 4) & 7);
  const long blk = id >> 7;                        // cb16 * nchunks + chunk
  const int chunk = (int)(blk % nchunks), cb = (int)(blk / nchunks);
  const int c = chunk * 8 + ci, co = cb * 16 + m;
  float g[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) g[tap] = (co < cout_pad) ? wp[((size_t)tap * cin_pad + c) * cout_pad + co] : 0.f;
  float t[4][3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const float g0 = g[kx], g1 = g[3 + kx], g2 = g[6 + kx];
    t[0][kx] = g0; t[1][kx] = 0.5f * ((g0 + g2) + g1); t[2][kx] = 0.5f * ((g0 + g2) - g1); t[3][kx] = g2;
  }
  float* u = wp + (size_t)9 * cin_pad * cout_pad + (size_t)blk * 2048 + ci * 16 + m;
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) {
    const float u0 = t[xi][0], u3 = t[xi][2];
    const float u1 = 0.5f * ((t[xi][0] + t[xi][2]) + t[xi][1]), u2 = 0.5f * ((t[xi][0] + t[xi][2]) - t[xi][1]);
    u[(xi * 4 + 0) * 128] = u0; u[(xi * 4 + 1) * 128] = u1; u[(xi * 4 + 2) * 128] = u2; u[(xi * 4 + 3) * 128] = u3;
  }
}
}  // namespace ms
