// 3x3 stride-1 convolutions with a handful of OUTPUT channels (<= 4) on the vector ALUs: the data-gradient that reaches the image
// (`inc.0`: 16 -> 1 channels at C2, 64 -> 3 at C4; encoder_decoder.py:441-445 backward).  On the matrix cores such a layer costs as much as a 16-column one
// (the MFMA tile is 16 wide: 54.8 us at 16 -> 1 @16x256x256, 15/16 of the products wasted); as 9*Cin*Cout FMAs per pixel it is a pure streaming kernel:
// read the gradient (and, with the BatchNorm-backward prologue, the raw conv output), write Cout planes.
//
// Workgroup = 256 threads = a 16 x 64 pixel tile, 4 consecutive pixels per thread; input channels are staged 4 at a time through LDS (18 x 72 window per
// channel, prologue al*g + be*u + de applied while staging, zeros outside the image AFTER the prologue = the conv's zero padding), double buffered; a thread
// reads its 3 x 6 window per channel as one ds_read_b128 + two ds_read_b32 per row.  Weights: the data-gradient packed layout [tap][cin_pad][cout_pad], read
// through uniform (scalar) loads.
#include <algorithm>
#include "ms_conv_kernel.h"
#include "maxstyle_hip.h"

namespace ms {

constexpr int kSmTH = 16, kSmTW = 64, kSmCK = 2, kSmIH = kSmTH + 2, kSmRS = kSmTW + 8, kSmPS = kSmIH * kSmRS;       // window columns x0-4 .. x0+67

template <int COUT, bool PRO2, typename AT = float>
__global__ __launch_bounds__(256) void conv3x3_small_cout_kernel(const void* __restrict__ in, const void* __restrict__ in2, void* __restrict__ out,
                                                                 const float* __restrict__ w, const float* __restrict__ pro_a, const float* __restrict__ pro_b,
                                                                 const float* __restrict__ pro_c, int pro_cstride, int Cin, int H, int W, int cin_pad, int cout_pad,
                                                                 int tiles_x) {
  __shared__ __attribute__((aligned(16))) float smem[2][kSmCK * kSmPS];
  const int tid = threadIdx.x;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int y0 = ty * kSmTH, x0 = tx * kSmTW;
  const size_t plane = (size_t)H * W;
  using IO = ActIO<AT>;                     // storage type of in / in2 / out: float | ms_bf16
  const size_t in_n = (size_t)n * Cin * plane;
  constexpr int ITEMS = kSmCK * kSmIH * (kSmRS / 4);        // float4 items per chunk: 4 x 18 x 18 = 1296
  constexpr int NI = (ITEMS + 255) / 256;
  float4 rg[NI], ru[PRO2 ? NI : 1];
  const int nchunks = (Cin + kSmCK - 1) / kSmCK;

  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int it = tid + j * 256;
      rg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (PRO2) ru[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < ITEMS) {
        const int f = it % (kSmRS / 4), row = it / (kSmRS / 4);
        const int r = row % kSmIH, c = row / kSmIH;
        const int Y = y0 - 1 + r, X = x0 - 4 + 4 * f, ci = c0 + c;
        if (ci < Cin && Y >= 0 && Y < H && X >= 0 && X < W) {          // W % 4 == 0: a quad is inside or outside as a whole
          const size_t off = in_n + (size_t)ci * plane + (size_t)Y * W + X;
          rg[j] = IO::ld4(in, off);
          if (PRO2) ru[j] = IO::ld4(in2, off);
        }
      }
    }
  };
  auto store_chunk = [&](float* buf, int c0) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int it = tid + j * 256;
      if (it < ITEMS) {
        const int f = it % (kSmRS / 4), row = it / (kSmRS / 4);
        const int r = row % kSmIH, c = row / kSmIH;
        const int Y = y0 - 1 + r, X = x0 - 4 + 4 * f, ci = c0 + c;
        float4 v = rg[j];
        if (PRO2) {
          const bool ok = (ci < Cin && Y >= 0 && Y < H && X >= 0 && X < W);
          if (ok) {
            const float al = pro_a[ci * pro_cstride], be = pro_b[ci * pro_cstride], de = pro_c[ci * pro_cstride];
            v.x = al * v.x + (be * ru[j].x + de); v.y = al * v.y + (be * ru[j].y + de);
            v.z = al * v.z + (be * ru[j].z + de); v.w = al * v.w + (be * ru[j].w + de);
          }
        }
        *reinterpret_cast<float4*>(buf + c * kSmPS + r * kSmRS + 4 * f) = v;
      }
    }
  };

  const int py = tid >> 4, px4 = (tid & 15) * 4;          // this thread's pixels: row y0+py, columns x0+px4 .. +3
  float acc[COUT][4];
#pragma unroll
  for (int o = 0; o < COUT; ++o)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[o][e] = 0.f;

  load_chunk(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    float* buf = smem[ch & 1];
    store_chunk(buf, ch * kSmCK);
    if (ch + 1 < nchunks) load_chunk((ch + 1) * kSmCK);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kSmCK; ++c) {
      const int ci = ch * kSmCK + c;
      if (ci < Cin) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* q = buf + c * kSmPS + (py + ky) * kSmRS + px4 + 3;       // window columns x-1 .. x+4 at LDS columns px4+3 .. px4+8
          const float4 mid = *reinterpret_cast<const float4*>(q + 1);
          const float win[6] = {q[0], mid.x, mid.y, mid.z, mid.w, q[5]};
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
              const float wv = w[((size_t)(ky * 3 + kx) * cin_pad + ci) * cout_pad + o];      // uniform address: scalar load
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[o][e] += wv * win[e + kx];
            }
        }
      }
    }
    // (the next iteration writes the OTHER buffer; the barrier of that iteration orders it against these reads)
  }
  const int y = y0 + py, x = x0 + px4;
  if (y < H && x < W) {
#pragma unroll
    for (int o = 0; o < COUT; ++o)
      IO::st4(out, ((size_t)n * COUT + o) * plane + (size_t)y * W + x, make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]));
  }
}

}  // namespace ms

using namespace ms;

extern "C" int ms_conv3x3_small_cout_ok(int Cout, int W) { return (Cout >= 1 && Cout <= 4 && W % 4 == 0) ? 1 : 0; }

// out [N,Cout,H,W] = conv3x3(P(in), w) with Cout <= 4, stride 1, padding 1; w = packed weights [9][cin_pad][cout_pad] (for a data-gradient: the
// data-gradient layout).  pro_mode 0: P = identity; 2: P = pro_a[c]*in + pro_b[c]*in2 + pro_c[c] (BatchNorm backward, coefficient records of stride pro_cstride).
template <typename AT>
static int small_cout_impl(const void* in, const void* in2, void* out, const float* w_packed, int N, int Cin, int H, int W, int Cout,
                           int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_cstride, void* stream) {
  if (N < 1 || Cin < 1 || H < 1 || W < 1 || !ms_conv3x3_small_cout_ok(Cout, W) || N > 65535) { set_error("ms_conv3x3_small_cout: Cout <= 4, W %% 4 == 0"); return MS_ERR_INVALID; }
  if (pro_mode != 0 && pro_mode != 2) { set_error("ms_conv3x3_small_cout: pro_mode 0 or 2"); return MS_ERR_INVALID; }
  if (pro_mode == 2 && (in2 == nullptr || pro_a == nullptr || pro_b == nullptr || pro_c == nullptr)) { set_error("ms_conv3x3_small_cout: prologue operands missing"); return MS_ERR_INVALID; }
  if (!aligned16(in) || !aligned16(out) || (in2 != nullptr && !aligned16(in2))) { set_error("ms_conv3x3_small_cout: tensors must be 16-byte aligned"); return MS_ERR_ALIGN; }
  const int tiles_x = cdiv(W, kSmTW), tiles_y = cdiv(H, kSmTH);
  const int cin_pad = (Cin + 3) / 4 * 4, cout_pad = (Cout + 63) / 64 * 64;
  dim3 grid(tiles_x * tiles_y, N), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int cs = pro_cstride < 1 ? 1 : pro_cstride;
#define MS_SM(CO, P2) MS_LAUNCH((conv3x3_small_cout_kernel<CO, P2, AT>), grid, block, 0, st, in, in2, out, w_packed, pro_a, pro_b, pro_c, cs, Cin, H, W, cin_pad, cout_pad, tiles_x)
  if (pro_mode == 2) { switch (Cout) { case 1: MS_SM(1, true); break; case 2: MS_SM(2, true); break; case 3: MS_SM(3, true); break; default: MS_SM(4, true); } }
  else { switch (Cout) { case 1: MS_SM(1, false); break; case 2: MS_SM(2, false); break; case 3: MS_SM(3, false); break; default: MS_SM(4, false); } }
#undef MS_SM
  return check_launch("conv3x3_small_cout");
}
extern "C" int ms_conv3x3_small_cout(const float* in, const float* in2, float* out, const float* w_packed, int N, int Cin, int H, int W, int Cout,
                                     int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_cstride, void* stream) {
  return small_cout_impl<float>(in, in2, out, w_packed, N, Cin, H, W, Cout, pro_mode, pro_a, pro_b, pro_c, pro_cstride, stream);
}
// bf16 activation storage (in, in2, out are bf16 bit patterns): see the bf16 section of include/maxstyle_hip.h
extern "C" int ms_conv3x3_small_cout_bf16(const uint16_t* in, const uint16_t* in2, uint16_t* out, const float* w_packed, int N, int Cin, int H, int W, int Cout,
                                          int pro_mode, const float* pro_a, const float* pro_b, const float* pro_c, int pro_cstride, void* stream) {
  return small_cout_impl<ms_bf16>(in, in2, out, w_packed, N, Cin, H, W, Cout, pro_mode, pro_a, pro_b, pro_c, pro_cstride, stream);
}
