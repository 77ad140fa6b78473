// Weight-gradient kernel template for gfx950 (exact-fp32 matrix cores, v_mfma_f32_16x16x4_f32).
// See ms_wgrad.hip for what it replaces in the reference and for the dispatch.
//
//   dW[m][n][ty][tx] = sum_{img, y, x}  P[img, m, y, x] * Q[img, n, S*y + ty - PAD, S*x + tx - PAD]
//
// i.e. a GEMM whose reduction dimension is the PIXEL index (millions long) and whose output is tiny.  Conv2d: P = gradient of the
// conv output, Q = conv input, dW = weight.grad [Cout][Cin][k][k].  ConvTranspose2d(k=2,s=2): P = its INPUT (low resolution),
// Q = gradient of its output, KS=2, S=2, PAD=0 -> dW = weight.grad [Cin][Cout][2][2].
//
// Persistent, wave-specialised workgroups (512 threads), same role split as the forward kernel (ms_conv_kernel.h):
//   waves 4-7  PRODUCERS  global --16-B loads--> registers --BatchNorm-backward prologue (P) / BatchNorm-apply + LeakyReLU prologue (Q)
//                         --> LDS tile (double buffered), one tile ahead of the consumers
//   waves 0-3  CONSUMERS  wave w owns row w of the 4-row pixel tile; per 16 pixels: ds_read_b32 fragments, KS*KS MFMAs per 16x16
//                         (m,n) block; the accumulators live in registers for the WHOLE run of tiles of the workgroup - no epilogue
//                         per tile.  At the end the four waves are summed through LDS and one partial dW block is written.
// MFMA operand mapping (16x16x4, A[m][k], B[k][n]): lane = (k = lane>>4, m|n = lane&15); step j of a 16-pixel segment multiplies
// pixel 4j+k, so a half-wave reads two ADJACENT pixels of 16 channel planes: plane strides == 2 (mod 32) make that conflict-free.
// A workgroup covers one (16*AB x 16*BB) channel block and a contiguous run of pixel tiles; a second kernel sums the partials of the
// workgroups that share a channel block in a fixed order (deterministic, no float atomics).
#pragma once
#include <algorithm>
#include "ms_common.h"

namespace ms {

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

struct WgArgs {
  const float* p; const float* p2; const float* q; float* partial;
  const float* pa; const float* pb; const float* pc; const float* qa; const float* qb;
  int N, M, Nq, Hp, Wp, Hq, Wq;       // Hq/Wq: STORED size of Q; logical size is 2x with q_ups
  int p_mode, q_mode, coef_stride; float slope;
  int tiles_x, tiles_y, ntiles, per, nslots, npairs_n;   // per: tiles per workgroup; npairs_n: channel blocks along n
};

template <int KS, int S, int AB, int BB, int TW, bool VEC, bool QUPS>
struct WgGeo {
  static constexpr int TH = 4;
  static constexpr int PAD = (KS == 3) ? 1 : 0;
  static constexpr int CO = (PAD == 0) ? 0 : (S == 1 ? 2 : 4);        // LDS column of logical column S*x0 (keeps 8-byte alignment of the interior)
  static constexpr int HALF = TW + (PAD ? 2 : 0);                      // S == 2: even / odd column planes
  static constexpr int RSQ = (S == 1) ? ((TW + KS - 1 - PAD + CO + 1) / 2 * 2) : 2 * HALF;
  static constexpr int QH = (TH - 1) * S + KS;
  static constexpr int QWL = (TW - 1) * S + KS;                        // logical columns a tile needs
  static constexpr int pad2(int v) { return v + ((2 - v % 32 + 32) % 32); }   // == 2 (mod 32)
  static constexpr int PSP = pad2(TH * TW);
  static constexpr int PSQ = pad2(QH * RSQ);
  static constexpr int BUF = 16 * AB * PSP + 16 * BB * PSQ;            // floats per stage buffer
  static constexpr int TAPS = KS * KS;
  static constexpr int VW = VEC ? 4 : 1;
  // staging items: P [16AB][TH][TW/VW];  Q interior [16BB][QH][S*TW/VW] (VEC) or all QWL columns (scalar); Q halo (VEC only)
  static constexpr int P_ROW = TW / VW;
  static constexpr int P_ITEMS = 16 * AB * TH * P_ROW;
  static constexpr int Q_ROW = VEC ? (S * TW / 4) : QWL;
  static constexpr int Q_ITEMS = 16 * BB * QH * Q_ROW;
  static constexpr int NHALO = VEC ? (QWL - S * TW) : 0;               // scalar columns outside the aligned interior: KS3/S1: 2 (left, right); KS3/S2: 1 (left)
  static constexpr int H_ITEMS = 16 * BB * QH * NHALO;
  static constexpr int NPI = (P_ITEMS + 255) / 256, NQI = (Q_ITEMS + 255) / 256, NHI = (H_ITEMS + 255) / 256;
  static constexpr int q_off(int lr, int col_rel) {                    // LDS offset inside a Q plane of logical (row, column) relative to the tile origin
    const int c = col_rel + CO;
    return lr * RSQ + ((S == 1) ? c : ((c & 1) * HALF + (c >> 1)));
  }
  static constexpr int tap_off(int tap) {                              // consumer: offset of tap (ky,kx) relative to pixel (row r -> S*r rows, column x)
    const int ky = tap / KS, kx = tap % KS;
    const int t = kx - PAD + CO;
    return ky * RSQ + ((S == 1) ? t : ((t & 1) * HALF + (t >> 1)));
  }
  static constexpr int RED = 2 * AB * BB * TAPS * 256;                 // floats of the end-of-run cross-wave reduction
  static constexpr int LDS_FLOATS = (2 * BUF > RED ? 2 * BUF : RED) + 16 * AB * 4 + 16 * BB * 4;
};

template <int KS, int S, int AB, int BB, int TW, bool VEC, bool QUPS>
__global__ __launch_bounds__(512, 2) void wgrad_mfma_kernel(const WgArgs a) {
  using G = WgGeo<KS, S, AB, BB, TW, VEC, QUPS>;
  constexpr int TH = G::TH, PAD = G::PAD, RSQ = G::RSQ, QH = G::QH, PSP = G::PSP, PSQ = G::PSQ, BUF = G::BUF, TAPS = G::TAPS;
  constexpr int VW = G::VW, NPI = G::NPI, NQI = G::NQI, NHI = G::NHI;
  static_assert(!QUPS || (KS == 3 && S == 1), "up-sampled fetch is for the 3x3 stride-1 convolution");
  static_assert(TW % 16 == 0, "tile width is a multiple of the 16-pixel MFMA segment");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* cfp = smem + (2 * BUF > G::RED ? 2 * BUF : G::RED);     // [16AB][4] P coefficients
  float* cfq = cfp + 16 * AB * 4;                                // [16BB][4] Q coefficients
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool producer = wave >= 4;
  // XCD-aware numbering (see ms_conv_kernel.h): consecutive virtual ids - the channel blocks of one pixel range, then the next range - share an L2
  const int vb = ((int)gridDim.x % 8 == 0) ? (((int)blockIdx.x % 8) * ((int)gridDim.x / 8) + (int)blockIdx.x / 8) : (int)blockIdx.x;
  const int npairs = (int)gridDim.x / a.nslots;
  const int pair = vb % npairs, slot = vb / npairs;
  const int m0 = (pair / a.npairs_n) * 16 * AB, n0 = (pair % a.npairs_n) * 16 * BB;
  const int t_begin = slot * a.per, t_end = min(a.ntiles, t_begin + a.per);
  const int T = t_end - t_begin;                                  // >= 1 by construction of the grid
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const int HqL = QUPS ? 2 * a.Hq : a.Hq, WqL = QUPS ? 2 * a.Wq : a.Wq;

  for (int c = threadIdx.x; c < 16 * AB; c += 512) {
    const int m = m0 + c;
    float x = 1.f, y = 0.f, z = 0.f;
    if (a.p_mode == 2 && m < a.M) { x = a.pa[m * a.coef_stride]; y = a.pb[m * a.coef_stride]; z = a.pc[m * a.coef_stride]; }
    cfp[c * 4] = x; cfp[c * 4 + 1] = y; cfp[c * 4 + 2] = z;
  }
  for (int c = threadIdx.x; c < 16 * BB; c += 512) {
    const int n = n0 + c;
    float x = 1.f, y = 0.f;
    if (a.q_mode == 1 && n < a.Nq) { x = a.qa[n * a.coef_stride]; y = a.qb[n * a.coef_stride]; }
    cfq[c * 4] = x; cfq[c * 4 + 1] = y;
  }

  if (producer) {
    // =========================================== PRODUCER waves ===========================================
    const int tid = threadIdx.x - 256;
    const size_t p_plane = (size_t)a.Hp * a.Wp, q_plane = (size_t)a.Hq * a.Wq;
    // tile-independent decode of the items (channel, row, column) and their LDS offsets
    int p_lds[NPI], p_rc[NPI];         // LDS float offset inside the P part (or -1), (row << 16) | column
    int q_lds[NQI], q_rc[NQI];         // LDS float offset of the item's first element (or -1), (row << 16) | (column_rel + 16)
    int h_lds[NHI > 0 ? NHI : 1], h_rc[NHI > 0 ? NHI : 1];
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
      const int it = tid + j * 256;
      p_lds[j] = -1; p_rc[j] = 0;
      if (it < G::P_ITEMS) {
        const int f = it % G::P_ROW, row = it / G::P_ROW;
        const int r = row % TH, c = row / TH;
        p_lds[j] = (c << 20) | (c * PSP + r * TW + f * VW);
        p_rc[j] = (r << 16) | (f * VW);
      }
    }
#pragma unroll
    for (int j = 0; j < NQI; ++j) {
      const int it = tid + j * 256;
      q_lds[j] = -1; q_rc[j] = 0;
      if (it < G::Q_ITEMS) {
        const int f = it % G::Q_ROW, row = it / G::Q_ROW;
        const int r = row % QH, c = row / QH;
        const int col_rel = VEC ? f * 4 : f - PAD;
        q_lds[j] = (c << 20) | (c * PSQ + G::q_off(r, col_rel));
        q_rc[j] = (r << 16) | (col_rel + 16);
      }
    }
    if constexpr (NHI > 0) {
#pragma unroll
      for (int j = 0; j < NHI; ++j) {
        const int it = tid + j * 256;
        h_lds[j] = -1; h_rc[j] = 0;
        if (it < G::H_ITEMS) {
          const int h = it % G::NHALO, row = it / G::NHALO;
          const int r = row % QH, c = row / QH;
          const int col_rel = (h == 0) ? -PAD : S * TW + (h - 1);      // left halo column(s) first, then the right ones
          h_lds[j] = (c << 20) | (c * PSQ + G::q_off(r, col_rel));
          h_rc[j] = (r << 16) | (col_rel + 16);
        }
      }
    }
    int p_goff[NPI], q_goff[NQI], h_goff[NHI > 0 ? NHI : 1];
    auto set_tile = [&](int tile, int& img) {
      const int tx = tile % a.tiles_x, t2 = tile / a.tiles_x;
      const int ty = t2 % a.tiles_y;
      img = t2 / a.tiles_y;
      const int y0 = ty * TH, x0 = tx * TW;
#pragma unroll
      for (int j = 0; j < NPI; ++j) {
        const int y = y0 + (p_rc[j] >> 16), x = x0 + (p_rc[j] & 0xFFFF);
        const bool ok = (p_lds[j] >= 0) && (y < a.Hp) && (x < a.Wp) && (m0 + (p_lds[j] >> 20) < a.M);
        p_goff[j] = ok ? (y * a.Wp + x) : -1;
      }
      auto q_addr = [&](int lds, int rc) {
        const int Y = y0 * S - PAD + (rc >> 16), X = x0 * S + (rc & 0xFFFF) - 16;
        const bool ok = (lds >= 0) && (Y >= 0) && (Y < HqL) && (X >= 0) && (X < WqL) && (n0 + (lds >> 20) < a.Nq);
        return ok ? ((QUPS ? (Y >> 1) : Y) * a.Wq + (QUPS ? (X >> 1) : X)) : -1;
      };
#pragma unroll
      for (int j = 0; j < NQI; ++j) q_goff[j] = q_addr(q_lds[j], q_rc[j]);
      if constexpr (NHI > 0) {
#pragma unroll
        for (int j = 0; j < NHI; ++j) h_goff[j] = q_addr(h_lds[j], h_rc[j]);
      }
    };
    float rp[NPI][VW], rp2[NPI][VW], rq[NQI][VW], rh[NHI > 0 ? NHI : 1];
    auto load_tile = [&](int img) {
      const float* pn = a.p + (size_t)img * a.M * p_plane;
      const float* p2n = (a.p_mode == 2) ? a.p2 + (size_t)img * a.M * p_plane : nullptr;
      const float* qn = a.q + (size_t)img * a.Nq * q_plane;
#pragma unroll
      for (int j = 0; j < NPI; ++j) {
        const bool ok = p_goff[j] >= 0;
        const size_t off = ok ? ((size_t)(m0 + (p_lds[j] >> 20)) * p_plane + (size_t)p_goff[j]) : 0;
        if constexpr (VEC) {
          const float4 v = ok ? *reinterpret_cast<const float4*>(pn + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          rp[j][0] = v.x; rp[j][1] = v.y; rp[j][2] = v.z; rp[j][3] = v.w;
          if (a.p_mode == 2) {
            const float4 u = ok ? *reinterpret_cast<const float4*>(p2n + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            rp2[j][0] = u.x; rp2[j][1] = u.y; rp2[j][2] = u.z; rp2[j][3] = u.w;
          }
        } else {
          rp[j][0] = ok ? pn[off] : 0.f;
          if (a.p_mode == 2) rp2[j][0] = ok ? p2n[off] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < NQI; ++j) {
        const bool ok = q_goff[j] >= 0;
        const size_t off = ok ? ((size_t)(n0 + (q_lds[j] >> 20)) * q_plane + (size_t)q_goff[j]) : 0;
        if constexpr (VEC && QUPS) {
          const float2 v = ok ? *reinterpret_cast<const float2*>(qn + off) : make_float2(0.f, 0.f);
          rq[j][0] = v.x; rq[j][1] = v.x; rq[j][2] = v.y; rq[j][3] = v.y;
        } else if constexpr (VEC) {
          const float4 v = ok ? *reinterpret_cast<const float4*>(qn + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          rq[j][0] = v.x; rq[j][1] = v.y; rq[j][2] = v.z; rq[j][3] = v.w;
        } else {
          rq[j][0] = ok ? qn[off] : 0.f;
        }
      }
      if constexpr (NHI > 0) {
#pragma unroll
        for (int j = 0; j < NHI; ++j) {
          const bool ok = h_goff[j] >= 0;
          rh[j] = ok ? qn[(size_t)(n0 + (h_lds[j] >> 20)) * q_plane + (size_t)h_goff[j]] : 0.f;
        }
      }
    };
    auto store_tile = [&](float* buf) {
      float* pl = buf;
      float* ql = buf + 16 * AB * PSP;
#pragma unroll
      for (int j = 0; j < NPI; ++j) {
        if (p_lds[j] < 0) continue;
        float v[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) v[e] = rp[j][e];
        if (a.p_mode == 2) {
          const int c = p_lds[j] >> 20;
          const float ca = cfp[c * 4], cb = cfp[c * 4 + 1], cc = cfp[c * 4 + 2];
#pragma unroll
          for (int e = 0; e < VW; ++e) v[e] = (p_goff[j] >= 0) ? (ca * v[e] + cb * rp2[j][e] + cc) : 0.f;     // pixels outside the image contribute nothing
        }
        float* dst = pl + (p_lds[j] & 0xFFFFF);
        if constexpr (VEC) {
          *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
          *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
        } else {
          dst[0] = v[0];
        }
      }
      auto q_pro = [&](float v, int c, bool ok) {
        if (a.q_mode == 1) return ok ? leaky(cfq[c * 4] * v + cfq[c * 4 + 1], a.slope) : 0.f;                // zero padding pads the ACTIVATED tensor
        return v;
      };
#pragma unroll
      for (int j = 0; j < NQI; ++j) {
        if (q_lds[j] < 0) continue;
        const int c = q_lds[j] >> 20;
        const bool ok = q_goff[j] >= 0;
        float v[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) v[e] = q_pro(rq[j][e], c, ok);
        float* dst = ql + (q_lds[j] & 0xFFFFF);
        if constexpr (VEC && S == 1) {
          *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
          *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
        } else if constexpr (VEC) {          // S == 2: even columns at dst, odd columns at dst + HALF  (column_rel % 4 == 0, CO even)
          *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[2]);
          *reinterpret_cast<float2*>(dst + G::HALF) = make_float2(v[1], v[3]);
        } else {
          dst[0] = v[0];
        }
      }
      if constexpr (NHI > 0) {
#pragma unroll
        for (int j = 0; j < NHI; ++j) {
          if (h_lds[j] < 0) continue;
          ql[h_lds[j] & 0xFFFFF] = q_pro(rh[j], h_lds[j] >> 20, h_goff[j] >= 0);
        }
      }
    };
    int img;
    set_tile(t_begin, img);
    load_tile(img);
    lds_barrier();                                    // barrier #0: coefficient tables visible
    for (int p = 0; p < T; ++p) {
      store_tile(smem + (p & 1) * BUF);               // uses the goff of tile p for the validity masks: before set_tile(p+1)
      if (p + 1 < T) { set_tile(t_begin + p + 1, img); load_tile(img); }
      lds_barrier();                                  // barrier #(p+1): tile p visible; consumers done with tile p-1
    }
  } else {
    // =========================================== CONSUMER waves ===========================================
    __builtin_amdgcn_s_setprio(2);
    const int mn = lane & 15, k = lane >> 4;
    wg_f32x4 acc[AB][BB][TAPS];
#pragma unroll
    for (int i = 0; i < AB; ++i)
#pragma unroll
      for (int j = 0; j < BB; ++j)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[i][j][t] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
    const int a_lane = mn * PSP + wave * TW + k;
    const int b_lane = 16 * AB * PSP + mn * PSQ + wave * S * RSQ + k;
    constexpr int SEG_UNROLL = (AB * BB == 4) ? 1 : TW / 16;
    auto compute = [&](const float* buf) {
      const float* ap = buf + a_lane;
      const float* bp = buf + b_lane;
      // one 16-pixel segment at a time for the 2x2-block tiles: unrolling across segments makes the scheduler hoist every LDS read
      // of the tile above the MFMAs and spill the accumulators
#pragma unroll SEG_UNROLL
      for (int seg = 0; seg < TW / 16; ++seg) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int x = seg * 16 + 4 * j;
          float af[AB];
#pragma unroll
          for (int i = 0; i < AB; ++i) af[i] = ap[i * 16 * PSP + x];
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            float bf[BB];
#pragma unroll
            for (int jb = 0; jb < BB; ++jb) bf[jb] = bp[jb * 16 * PSQ + G::tap_off(t) + x];
#pragma unroll
            for (int i = 0; i < AB; ++i)
#pragma unroll
              for (int jb = 0; jb < BB; ++jb) acc[i][jb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[jb], acc[i][jb][t], 0, 0, 0);
          }
        }
      }
    };
    lds_barrier();                                      // barrier #0
    lds_barrier();                                      // barrier #1: tile 0 is in buffer 0
    for (int p = 0; p < T; ++p) {
      compute(smem + (p & 1) * BUF);
      if (p + 1 < T) lds_barrier();                     // barrier #(p+2)
    }
    // ---- sum the four waves (each holds the partial of its tile row): waves 2,3 -> LDS -> waves 0,1 ; wave 1 -> LDS -> wave 0 ----
    lds_barrier();                                      // R0: every wave is done reading the stage buffers
    float* red = smem;
    auto dump = [&](int half) {
#pragma unroll
      for (int i = 0; i < AB; ++i)
#pragma unroll
        for (int j = 0; j < BB; ++j)
#pragma unroll
          for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((half * AB * BB + i * BB + j) * TAPS + t) * 256 + r * 64 + lane] = acc[i][j][t][r];
    };
    auto absorb = [&](int half) {
#pragma unroll
      for (int i = 0; i < AB; ++i)
#pragma unroll
        for (int j = 0; j < BB; ++j)
#pragma unroll
          for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][t][r] += red[((half * AB * BB + i * BB + j) * TAPS + t) * 256 + r * 64 + lane];
    };
    if (wave >= 2) dump(wave - 2);
    lds_barrier();                                      // R1
    if (wave < 2) absorb(wave);
    lds_barrier();                                      // R2: waves 0,1 have read; the area may be rewritten
    if (wave == 1) dump(0);
    lds_barrier();                                      // R3
    if (wave == 0) {
      absorb(0);
      // D layout: column (n) = lane & 15, rows (m) = 4*(lane>>4) + reg.  partial[slot][m][n][tap] in weight.grad layout
      float* out = a.partial + (size_t)slot * a.M * a.Nq * TAPS;
#pragma unroll
      for (int i = 0; i < AB; ++i)
#pragma unroll
        for (int j = 0; j < BB; ++j) {
          const int n = n0 + j * 16 + mn;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = m0 + i * 16 + 4 * k + r;
            if (m < a.M && n < a.Nq) {
#pragma unroll
              for (int t = 0; t < TAPS; ++t) out[((size_t)m * a.Nq + n) * TAPS + t] = acc[i][j][t][r];
            }
          }
        }
    }
    return;
  }
  // producers: match the consumers' end-of-run barriers R0..R3
  lds_barrier(); lds_barrier(); lds_barrier(); lds_barrier();
}

template <int KS, int S, int AB, int BB, int TW, bool VEC, bool QUPS>
int launch_wgrad(WgArgs a, int max_wg, hipStream_t st) {
  using G = WgGeo<KS, S, AB, BB, TW, VEC, QUPS>;
  const size_t lds_bytes = sizeof(float) * (size_t)G::LDS_FLOATS;
  static_assert(sizeof(float) * (size_t)G::LDS_FLOATS <= 160 * 1024, "tile does not fit the LDS");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)wgrad_mfma_kernel<KS, S, AB, BB, TW, VEC, QUPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    attr_set = true;
  }
  a.tiles_x = cdiv(a.Wp, TW); a.tiles_y = cdiv(a.Hp, G::TH);
  a.ntiles = a.N * a.tiles_x * a.tiles_y;
  const int npm = cdiv(a.M, 16 * AB);
  a.npairs_n = cdiv(a.Nq, 16 * BB);
  const int npairs = npm * a.npairs_n;
  int nslots = std::max(1, std::min(a.ntiles, max_wg / npairs));
  a.per = cdiv(a.ntiles, nslots);
  nslots = cdiv(a.ntiles, a.per);                      // no empty workgroup
  a.nslots = nslots;
  MS_LAUNCH((wgrad_mfma_kernel<KS, S, AB, BB, TW, VEC, QUPS>), dim3((unsigned)(npairs * nslots)), dim3(512), lds_bytes, st, a);
  return check_launch("wgrad_mfma");
}

// number of partial slots launch_wgrad will use (the caller sizes the workspace with it)
template <int AB, int BB, int TW>
int wgrad_slots(int N, int M, int Nq, int Hp, int Wp, int max_wg) {
  const int ntiles = N * cdiv(Wp, TW) * cdiv(Hp, 4);
  const int npairs = cdiv(M, 16 * AB) * cdiv(Nq, 16 * BB);
  const int nslots = std::max(1, std::min(ntiles, max_wg / npairs));
  const int per = cdiv(ntiles, nslots);
  return cdiv(ntiles, per);
}

}  // namespace ms
