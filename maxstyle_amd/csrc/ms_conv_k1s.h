// Streaming form of the 1x1 convolutions (the residual blocks' skip paths `conv_input` and their data-gradients, encoder_decoder.py:62-64, 344-346) for gfx950.
//
// A 1x1 convolution over [N, Cin, H, W] is a GEMM whose pixel dimension has no structure: out[p, co] = sum_c x[p, c] W[c, co].  With <= 128 input channels it is
// HBM-bound (config 4, 64 -> 64 @16x320x320 with the residual tail: 1.26 GB for 13 GFLOP), and the first-generation kernel (ms_conv_kernel.h: 8-channel chunks through
// LDS, ONE chunk in flight per workgroup behind a workgroup barrier) keeps ~16 KB per CU in flight: 2.7 TB/s (profiles/r04_experiments.txt 8).  Here every wave streams on
// its own, with no LDS and no barrier on the activation path:
//   * a unit = 64 consecutive pixels of one image plane; lane (m, k) loads 16 bytes = pixels 4m .. 4m+3 of channel 4g + k: ONE buffer_load_dwordx4 per wave and 4-channel
//     group g is the A fragment of four M-tiles at once (M-tile t = pixels {4i + t});
//   * a ring of D = 16 such loads per wave is kept in flight across units (15 KB per wave, ~120 KB per CU), the MFMAs of group g run while group g + 15 is fetched;
//   * the weight slice [Cin][16 NT] sits in LDS for the life of the workgroup (B fragments: one ds_read_b32 per MFMA column block);
//   * M row 4k + r of tile t is pixel 4k + 16r + t of the unit (lane m loads its 16 bytes at pixel 4 (m >> 2) + 16 (m & 3)): in the accumulator layout the four k-lanes of
//     an output channel then hold ADJACENT pixel quads, so every 16-byte store / u load instruction covers 64 contiguous bytes per channel (with the natural mapping,
//     pixel 4m + t, the quads of a store sat 64 bytes apart: 0.52 -> 0.61 of HBM on the dominant shape, profiles/r04_experiments.txt 17); the epilogue (plain + bias, or the residual tail
//     out = lrelu((sc u + sh) + (acc + bias)) at the same or at twice the resolution) reads / writes 16-byte pieces, 64 contiguous bytes per lane.
// The channels are accumulated in the first generation's order (g ascending, one v_mfma_f32_16x16x4_f32 per group): same bits.
// Carries the same side jobs as the first generation: the rider (ConvArgs::ride_*) and the cross-workgroup finalize of a residual tail (xf_*, epilogue kind).
// Eligibility (conv_k1s_eligible): fp32 storage, no prologue, Cin a power of two in 16 .. 128, H W % 4 == 0, plain / ConvTranspose / same-resolution residual-tail
// epilogue, no statistics, at least four 64-pixel units per CU.
#pragma once
#include "ms_conv_kernel.h"

namespace ms {

#ifndef MS_K1S_D
#define MS_K1S_D 16
#endif
constexpr int kK1sD = MS_K1S_D;              // loads in flight per wave (8: measured no better, profiles/r04_experiments.txt 16)
constexpr int kK1sOob = (int)0x80000000;

// EPI: 0 plain | 2 ConvTranspose2d(k=2, s=2) as a GEMM with 4 Cout columns (NT = 4: column block q = (dy, dx) of 16 output channels; out[n, co, 2y+dy, 2x+dx]) |
//      4 residual tail | 5 residual tail at twice the resolution; NCGS = min(Cin / 4, D)
template <int NT, int EPI, int NCGS>
__global__ __launch_bounds__(256, 2) void conv_k1s_kernel(const ConvArgs a) {
  constexpr int D = kK1sD, COUT_TILE = 16 * NT, WS = (NT == 1) ? 16 : 16 * NT + 16;
  typedef unsigned lu32x4_t __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];                 // [Cin][WS]
  const int wave = __builtin_amdgcn_readfirstlane(MS_TID >> 6), lane = MS_TID & 63, m = lane & 15, k = lane >> 4;      // (wave index in an SGPR: every unit / offset below is wave-uniform)
  const int ncb = a.ncb;
  const int HW = a.Hs * a.Ws;
  const int upi = (HW + 63) >> 6, U = a.N * upi;
  const int NCG = a.cin_pad >> 2, lg = 31 - __builtin_clz(NCG);
  const int cb = (int)blockIdx.x % ncb, wg = (int)blockIdx.x / ncb, S = (int)gridDim.x / ncb;
  const int first = wg * 4 + wave, stride = S * 4;
  const int my_units = (first < U) ? (U - first + stride - 1) / stride : 0;
  const int total = my_units << lg;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((unsigned)a.N * a.Cin * HW * 4u), 0x00020000);

  // ---- load cursor: the unit / channel group the next issued load belongs to ----
  int l_u = first, l_cg = 0, l_soff = 0, l_voff = kK1sOob;
  auto l_set = [&]() {
    const int n = l_u / upi, p0 = (l_u - n * upi) << 6, px = p0 + 4 * (m >> 2) + 16 * (m & 3);      // (M row m = 4k + r of tile t <-> pixel 4k + 16r + t: see the epilogue)
    l_voff = (l_u < U && px < HW) ? ((k * HW + px) << 2) : kK1sOob;
    l_soff = (n * a.Cin * HW) << 2;
  };
  auto issue = [&]() -> float4 {
    const lu32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, l_voff, l_soff + ((l_cg * HW) << 4), 0);
    if (++l_cg == NCG) { l_cg = 0; l_u += stride; l_set(); }
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  };
  float4 ring[D];
  l_set();
#pragma unroll
  for (int j = 0; j < D; ++j) ring[j] = (j < total) ? issue() : make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- side jobs, while the first loads fly ----
  if (a.ride_out != nullptr)
    for (int c = (int)blockIdx.x * 4 + wave; c < a.ride_C; c += 4 * (int)gridDim.x) conv_ride(a, c, lane);
  unsigned xf_tag = 0u;
  int xf_nparts = 0;
  const bool xf_epi = (EPI != 0) && (a.xf_tab != nullptr);
  if (xf_epi) xfin_header(a, xf_tag, xf_nparts);
  // weight slice -> LDS
  {
    constexpr int Q = COUT_TILE / 4;
    for (int idx = MS_TID; idx < a.cin_pad * Q; idx += 256) {
      const int c = idx / Q, j4 = idx - c * Q;
      // EPI 2: the slice's column block q holds GEMM columns q * Cout + 16 cb .. + 15 (the four (dy, dx) positions of this item's 16 output channels)
      const size_t src = (EPI == 2) ? ((size_t)c * a.cout_pad + (size_t)(j4 >> 2) * a.cout_real + cb * 16 + 4 * (j4 & 3)) : ((size_t)c * a.cout_pad + cb * COUT_TILE + 4 * j4);
      *reinterpret_cast<float4*>(smem + c * WS + 4 * j4) = *reinterpret_cast<const float4*>(a.w + src);
    }
  }
  __syncthreads();
  if (xf_epi) xfin_produce(a, xf_tag, xf_nparts);

  // ---- per-lane epilogue constants ----
  float bias_v[NT], mk_sc[NT], mk_sh[NT];
  bool co_ok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = (EPI == 2) ? (cb * 16 + m) : (cb * COUT_TILE + j * 16 + m);
    co_ok[j] = co < ((EPI == 2) ? a.cout_real : a.Cout);
    bias_v[j] = (a.bias != nullptr && co_ok[j]) ? a.bias[co] : 0.f;
    mk_sc[j] = mk_sh[j] = 0.f;
    if (EPI != 0 && !xf_epi) {
      const float4 cf = co_ok[j] ? reinterpret_cast<const float4*>(a.mk_coef)[co] : make_float4(0.f, 0.f, 0.f, 0.f);
      mk_sc[j] = cf.x; mk_sh[j] = cf.y;
    }
  }
  if (xf_epi) {
    // (scale, shift) of this lane's channels, published by some wave of this launch a few microseconds from now; the first epilogue is only Cin / 4 MFMA groups
    // away, so the poll sits here, under the ring's first loads, rather than inside the stream
    conv_u64_t g[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j) xfin_peek(a, min(cb * COUT_TILE + j * 16 + m, a.xf_C - 1), (int)blockIdx.x & (kXfinRep - 1), g[j]);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float2 cf = co_ok[j] ? xfin_poll(a, cb * COUT_TILE + j * 16 + m, (int)blockIdx.x & (kXfinRep - 1), xf_tag, g[j]) : make_float2(0.f, 0.f);
      mk_sc[j] = cf.x; mk_sh[j] = cf.y;
    }
  }

  f32x4 acc[4][NT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // D layout of a 16x16 MFMA tile: this lane holds output channel m of rows 4k + r of M-tile t, i.e. pixels p0 + 4k + 16r + t: quad r of the lane at pixel 4k + 16r
  // EPI 4 with Cin >= 64: the residual tail's u values of the unit being accumulated, requested at the head of the unit's LAST ring round (16 MFMA groups ahead of
  // their use) - every load of the epilogue in one batch and in front of its stores (which the compiler must assume to alias the later loads)
  float4 uu[(EPI == 4) ? NT : 1][4];
  auto fetch_u = [&](int u) {
    const int n = u / upi, p0 = (u - n * upi) << 6, pl = p0 + 4 * k;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const size_t pb = ((size_t)n * a.Cout + cb * COUT_TILE + j * 16 + m) * (size_t)HW;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        uu[(EPI == 4) ? j : 0][r] = (co_ok[j] && pl + 16 * r < HW) ? *reinterpret_cast<const float4*>(a.mk_u + pb + pl + 16 * r) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto epilogue = [&](int u) {
    const int n = u / upi, p0 = (u - n * upi) << 6;
    const int pl = p0 + 4 * k;                                    // this lane's first pixel quad; quad r sits 16 r pixels on: the four k-lanes of a channel write 64 contiguous bytes per store
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][j][r] += bias_v[j];
    if (EPI == 2) {
      // pixel (y, x), channel co, position (dy, dx) -> out[n, co, 2y + dy, 2x + dx]: the lane's four pixels x .. x+3 and both dx are 8 consecutive floats of row 2y + dy
      int y = pl / a.Ws, x = pl - y * a.Ws;
      const int Wo = 2 * a.Ws;
      const size_t pb = ((size_t)n * a.cout_real + cb * 16 + m) * (size_t)(4 * HW);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (co_ok[0] && pl + 16 * r < HW) {
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            const size_t off = pb + (size_t)(2 * y + dy) * Wo + 2 * x;
            *reinterpret_cast<float4*>(a.out + off) = make_float4(acc[0][2 * dy][r], acc[0][2 * dy + 1][r], acc[1][2 * dy][r], acc[1][2 * dy + 1][r]);
            *reinterpret_cast<float4*>(a.out + off + 4) = make_float4(acc[2][2 * dy][r], acc[2][2 * dy + 1][r], acc[3][2 * dy][r], acc[3][2 * dy + 1][r]);
          }
        }
        x += 16;
        while (x >= a.Ws) { x -= a.Ws; ++y; }
      }
    } else if (EPI == 5) {
      // every value feeds a 2 x 2 block of outputs: (y, x) of the lane's pixel quads (a quad never crosses a row: Ws % 4 == 0)
      int y = pl / a.Ws, x = pl - y * a.Ws;
      const int Wo = 2 * a.Ws;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (pl + 16 * r < HW) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (!co_ok[j]) continue;
            const float sc = mk_sc[j], sh = mk_sh[j];
            const size_t pb = ((size_t)n * a.Cout + cb * COUT_TILE + j * 16 + m) * (size_t)(4 * HW);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
              const size_t off = pb + (size_t)(2 * y + dy) * Wo + 2 * x;
              const float4 t0 = *reinterpret_cast<const float4*>(a.mk_u + off), t1 = *reinterpret_cast<const float4*>(a.mk_u + off + 4);
              float4 o0, o1;
              o0.x = leaky((sc * t0.x + sh) + acc[0][j][r], a.mk_slope); o0.y = leaky((sc * t0.y + sh) + acc[0][j][r], a.mk_slope);
              o0.z = leaky((sc * t0.z + sh) + acc[1][j][r], a.mk_slope); o0.w = leaky((sc * t0.w + sh) + acc[1][j][r], a.mk_slope);
              o1.x = leaky((sc * t1.x + sh) + acc[2][j][r], a.mk_slope); o1.y = leaky((sc * t1.y + sh) + acc[2][j][r], a.mk_slope);
              o1.z = leaky((sc * t1.z + sh) + acc[3][j][r], a.mk_slope); o1.w = leaky((sc * t1.w + sh) + acc[3][j][r], a.mk_slope);
              *reinterpret_cast<float4*>(a.out + off) = o0;
              *reinterpret_cast<float4*>(a.out + off + 4) = o1;
            }
          }
        }
        x += 16;
        while (x >= a.Ws) { x -= a.Ws; ++y; }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (!co_ok[j]) continue;
        const size_t pb = ((size_t)n * a.Cout + cb * COUT_TILE + j * 16 + m) * (size_t)HW;
        const float sc = mk_sc[j], sh = mk_sh[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (pl + 16 * r >= HW) continue;
          float4 o = make_float4(acc[0][j][r], acc[1][j][r], acc[2][j][r], acc[3][j][r]);
          if (EPI == 4) {
            const float4 t = uu[(EPI == 4) ? j : 0][r];
            o.x = leaky((sc * t.x + sh) + o.x, a.mk_slope); o.y = leaky((sc * t.y + sh) + o.y, a.mk_slope);
            o.z = leaky((sc * t.z + sh) + o.z, a.mk_slope); o.w = leaky((sc * t.w + sh) + o.w, a.mk_slope);
          }
          *reinterpret_cast<float4*>(a.out + pb + pl + 16 * r) = o;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- the stream: position s = (unit index) * NCG + channel group; ring slot s % D ----
  const int b_lane = k * WS + m;
  int c_u = first;                                       // unit being accumulated
  for (int base = 0; base < total; base += D) {
    const int cg0 = base & (NCG - 1);
    if (EPI == 4 && NCGS == D && ((base + D) & (NCG - 1)) == 0) fetch_u(c_u);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if (NCGS < D && (j % NCGS) == 0 && base + j >= total) break;          // (NCG < D: the stream may end inside a body, at a unit boundary)
      const int cg = (NCGS < D) ? (j % NCGS) : (cg0 + j);
      const float4 av = ring[j];
      float bf[NT];
#pragma unroll
      for (int q = 0; q < NT; ++q) bf[q] = smem[b_lane + cg * 4 * WS + 16 * q];
      const float ac[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[t][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[t], bf[q], acc[t][q], 0, 0, 0);
      if (base + j + D < total) ring[j] = issue();
      const bool unit_end = (NCGS < D) ? ((j % NCGS) == NCGS - 1) : (j == D - 1 && ((base + D) & (NCG - 1)) == 0);
      if (unit_end) { if (EPI == 4 && NCGS < D) fetch_u(c_u); epilogue(c_u); c_u += stride; }
    }
  }
}

int conv_k1s_switch();      // ms_conv.hip: option "conv.k1s" (0: off - A/B runs and the same-bits tests)
inline bool conv_k1s_eligible(const ConvArgs& a, int ks, int stride, int fetch) {
  if (conv_k1s_switch() == 0 || ks != 1 || stride != 1 || fetch != FETCH_NORMAL || a.act_bf16 != 0 || a.pro_mode != 0 || a.in2 != nullptr || a.stats != nullptr ||
      !(a.epi_mode == 0 || a.epi_mode == 2 || a.epi_mode == 4 || a.epi_mode == 5)) return false;
  if (a.epi_mode == 2 && (a.cout_real % 16 != 0 || a.Ws % 4 != 0 || a.xf_tab != nullptr)) return false;
  if (a.xf_tab != nullptr && (a.epi_mode == 0 || a.epi_mode == 2)) return false;                     // (a prologue-kind `_xfin`: there is no prologue here)
  const int C = a.Cin;
  // measured (tools/ab_k1.py, profiles/r04_experiments.txt 8, 17, 19): with the 64-byte store mapping the stream also beats the tiled kernel on config 2's 16 / 32-channel
  // top levels (16 -> 16 @16x256x256 tail 39.8 -> 34.7 us = 0.73 of HBM); the half-resolution tails keep the tiled kernel
  constexpr int cmin = 16;
  if (C < cmin || C > 128 || (C & (C - 1)) != 0 || a.epi_mode == 5) return false;
  const long HW = (long)a.Hs * a.Ws;
  if (HW % 4 != 0 || (a.epi_mode == 5 && a.Ws % 4 != 0)) return false;
  if (a.epi_mode == 5 && C == 16 && a.Cout > 32) return false;                // (the one instantiation whose ring does not stay in registers; no layer has this shape)
  if ((long long)a.N * C * HW * 4 >= (1LL << 31)) return false;
  if (!aligned16(a.in) || !aligned16(a.out) || !aligned16(a.w) || (a.epi_mode != 0 && !aligned16(a.mk_u))) return false;
  // a streaming kernel: it pays where there is a stream - at least one 4-wave workgroup of 64-pixel units per CU (config 2's 64 / 128-channel layers on 32 x 32 images
  // have a quarter of that and lose: 13.4 -> 18.3 us; the small, channel-heavy levels keep the tiled kernel)
  return (long)a.N * ((HW + 63) / 64) * cdiv(a.Cout, 64) >= 4L * num_cus();      // (EPI 2: a.Cout = the 4 Cout GEMM columns = 64 per item)
}

template <int NT, int EPI, int NCGS>
int launch_conv_k1s_t(ConvArgs a, hipStream_t st) {
  constexpr int WS = (NT == 1) ? 16 : 16 * NT + 16;
  const size_t lds_bytes = sizeof(float) * (size_t)a.cin_pad * WS;
  a.ncb = (EPI == 2) ? a.cout_real / 16 : cdiv(a.Cout, 16 * NT);
  static std::mutex mu;
  static std::map<size_t, int> occ;
  int per_cu;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = occ.find(lds_bytes);
    if (it == occ.end()) {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv_k1s_kernel<NT, EPI, NCGS>, 256, lds_bytes) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
      it = occ.emplace(lds_bytes, n).first;
    }
    per_cu = it->second;
  }
  const long U = (long)a.N * (((long)a.Hs * a.Ws + 63) / 64);
  long nblocks = std::min<long>(cdiv(U, 4L) * a.ncb, (long)num_cus() * per_cu);
  if (nblocks > a.ncb) nblocks -= nblocks % a.ncb;
  if (nblocks < a.ncb) nblocks = a.ncb;
  MS_LAUNCH((conv_k1s_kernel<NT, EPI, NCGS>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, a);
  return check_launch("conv_k1s");
}

int conv_dispatch_k1s(const ConvArgs& a, hipStream_t st);      // ms_conv_inst_k1s.hip

}  // namespace ms
