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
#include <mutex>
#include <type_traits>
#include "ms_common.h"

namespace ms {

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

struct WgArgs {
  const float* p; const float* p2; const float* q; float* partial;
  const float* pa; const float* pb; const float* pc; const float* qa; const float* qb;
  int N, M, Nq, Hp, Wp, Hq, Wq;       // Hq/Wq: STORED size of Q; logical size is 2x with q_ups
  int p_mode, q_mode, coef_stride; float slope;
  int tiles_x, tiles_y, ntiles, per, nslots, npairs_n;   // per: tiles per workgroup; npairs_n: channel blocks along n
  long long* trace;                   // debug: per-tile cycle stamps of workgroup 0 (wave 0 = consumer, wave 4 = producer), or null
  int dbg;                            // timing-only ablation bits (MS_WGRAD_DBG): 1 skip the MFMA loop, 2 skip the global loads, 4 skip the LDS stores
};

template <int KS, int S, int AB, int BB, int TW, bool VEC, bool QUPS>
struct WgGeo {
  static constexpr int TH = 4;
  static constexpr int PAD = (KS == 3) ? 1 : 0;
  // WIDE (stride 1): the MFMA k index is BLOCKED - lane group k of a 16-pixel segment owns pixels 4k..4k+3 - so a lane reads its four A
  // values with one ds_read_b128 and the six B values of a kernel row (pixels 4k-1..4k+4) with one ds_read_b128 + one ds_read_b64: 28
  // LDS-array cycles per wave and segment instead of 80 with one ds_read_b32 per operand, which had the LDS pipe saturated (4 consumer
  // waves x 40 reads x 2 cycles per 288-cycle MFMA group = 111 %).  Plane strides == 8 (mod 16) floats keep the b128 reads conflict-free
  // (lane groups of ds_read_b128: MI355X_MICROARCH.md LDS table).  Stride 2 keeps the interleaved k index and ds_read_b32 (few FLOPs there).
  static constexpr bool WIDE = (S == 1);
  static constexpr int CO = (PAD == 0) ? 0 : (S == 1 ? 1 : 4);        // LDS column of logical column S*x0 (S 1: the left halo sits at column 0)
  static constexpr int HALF = TW + (PAD ? 2 : 0);                      // S == 2: even / odd column planes
  static constexpr int RSQ = (S == 1) ? ((TW + KS - 1 + 3) / 4 * 4) : 2 * HALF;
  static constexpr int QH = (TH - 1) * S + KS;
  static constexpr int QWL = (TW - 1) * S + KS;                        // logical columns a tile needs
  static constexpr int pad2(int v) { return WIDE ? v + ((8 - v % 16 + 16) % 16) : v + ((2 - v % 32 + 32) % 32); }   // == 8 (mod 16) | == 2 (mod 32)
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
  static constexpr int LDS_FLOATS = (2 * BUF > RED ? 2 * BUF : RED);
};

template <int KS, int S, int AB, int BB, int TW, bool VEC, bool QUPS, int PM, int QM>
__global__ __launch_bounds__(512, 2) void wgrad_mfma_kernel(const WgArgs a) {
  using G = WgGeo<KS, S, AB, BB, TW, VEC, QUPS>;
  constexpr int TH = G::TH, PAD = G::PAD, RSQ = G::RSQ, QH = G::QH, PSP = G::PSP, PSQ = G::PSQ, BUF = G::BUF, TAPS = G::TAPS;
  constexpr int VW = G::VW, NPI = G::NPI, NQI = G::NQI, NHI = G::NHI;
  static_assert(!QUPS || (KS == 3 && S == 1), "up-sampled fetch is for the 3x3 stride-1 convolution");
  static_assert(TW % 16 == 0, "tile width is a multiple of the 16-pixel MFMA segment");
  static_assert((PM == 0 || PM == 2) && (QM == 0 || QM == 1), "prologue modes");
  extern __shared__ __attribute__((aligned(16))) float smem[];
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

  if (producer) {
    // =========================================== PRODUCER waves ===========================================
    // Every VALU instruction here competes with the consumers' MFMA issue on the same SIMD (measured with in-kernel cycle stamps: a
    // staging pass of ~400 VALU instructions took 7 k cycles under a saturated matrix pipe, and the consumers waited for it).  So the
    // steady-state path is stripped to loads, the prologue FMAs and LDS stores: everything that does not depend on the tile is hoisted
    // into registers (LDS offsets, image-relative element offsets, per-channel coefficients), masks and zero-fill selects only run for
    // tiles that touch the image border (wave-uniform branch), and two register sets keep two tiles of loads in flight.
    __builtin_amdgcn_s_setprio(3);      // staging waves win issue arbitration (see ms_conv_wide.h)
    const int tid = threadIdx.x - 256;
    const int p_plane = a.Hp * a.Wp, q_plane = a.Hq * a.Wq;       // host checks that one image of P / Q stays below 2^31 elements
    typedef unsigned long long mask_t;
    int p_lds[NPI], p_rc[NPI], p_off[NPI];         // LDS float offset (or -1: no item); (row << 16) | column; c*plane + r*Wp + col (relative to the tile origin)
    int q_lds[NQI], q_rc[NQI], q_off[NQI];         // (row << 16) | (column_rel + 16)
    int h_lds[NHI > 0 ? NHI : 1], h_rc[NHI > 0 ? NHI : 1], h_off[NHI > 0 ? NHI : 1];
    float p_ca[PM == 2 ? NPI : 1], p_cb[PM == 2 ? NPI : 1], p_cc[PM == 2 ? NPI : 1];
    float q_ca[QM == 1 ? NQI : 1], q_cb[QM == 1 ? NQI : 1], h_ca[(QM == 1 && NHI > 0) ? NHI : 1], h_cb[(QM == 1 && NHI > 0) ? NHI : 1];
    mask_t p_chan = 0, q_chan = 0, h_chan = 0;     // bit j: item j exists and its channel is a real one
    mask_t p_all = 0, q_all = 0;                   // bit j: item j exists
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
      const int it = tid + j * 256;
      p_lds[j] = -1; p_rc[j] = 0; p_off[j] = 0;
      if (PM == 2) { p_ca[j] = 1.f; p_cb[j] = 0.f; p_cc[j] = 0.f; }
      if (it < G::P_ITEMS) {
        const int f = it % G::P_ROW, row = it / G::P_ROW;
        const int r = row % TH, c = row / TH;
        p_lds[j] = c * PSP + r * TW + f * VW;
        p_rc[j] = (r << 16) | (f * VW);
        p_all |= (mask_t)1 << j;
        if (m0 + c < a.M) {
          p_chan |= (mask_t)1 << j;
          p_off[j] = (m0 + c) * p_plane + r * a.Wp + f * VW;
          if (PM == 2) { const int ci = (m0 + c) * a.coef_stride; p_ca[j] = a.pa[ci]; p_cb[j] = a.pb[ci]; p_cc[j] = a.pc[ci]; }
        }
      }
    }
    auto q_item = [&](int c, int r, int col_rel, int& lds, int& rc, int& off, float& ca, float& cb, mask_t& chan, int j) {
      lds = c * PSQ + G::q_off(r, col_rel);
      rc = (r << 16) | (col_rel + 16);
      if (n0 + c < a.Nq) {
        chan |= (mask_t)1 << j;
        // stored offset relative to the tile origin; with the up-sampled fetch the logical (row, column) maps to (row >> 1, column >> 1)
        // and tile origins are even, so the shift distributes (arithmetic shift: floor for the -1 halo)
        const int ry = QUPS ? ((r - PAD) >> 1) : (r - PAD), rx = QUPS ? (col_rel >> 1) : col_rel;
        off = (n0 + c) * q_plane + ry * a.Wq + rx;
        if (QM == 1) { ca = a.qa[(n0 + c) * a.coef_stride]; cb = a.qb[(n0 + c) * a.coef_stride]; }
      }
    };
#pragma unroll
    for (int j = 0; j < NQI; ++j) {
      const int it = tid + j * 256;
      q_lds[j] = -1; q_rc[j] = 0; q_off[j] = 0;
      float ca = 1.f, cb = 0.f;
      if (it < G::Q_ITEMS) {
        const int f = it % G::Q_ROW, row = it / G::Q_ROW;
        q_all |= (mask_t)1 << j;
        q_item(row / QH, row % QH, VEC ? f * 4 : f - PAD, q_lds[j], q_rc[j], q_off[j], ca, cb, q_chan, j);
      }
      if (QM == 1) { q_ca[j] = ca; q_cb[j] = cb; }
    }
    if constexpr (NHI > 0) {
#pragma unroll
      for (int j = 0; j < NHI; ++j) {
        const int it = tid + j * 256;
        h_lds[j] = -1; h_rc[j] = 0; h_off[j] = 0;
        float ca = 1.f, cb = 0.f;
        if (it < G::H_ITEMS) {
          const int h = it % G::NHALO, row = it / G::NHALO;
          q_item(row / QH, row % QH, (h == 0) ? -PAD : S * TW + (h - 1), h_lds[j], h_rc[j], h_off[j], ca, cb, h_chan, j);     // left halo column(s) first, then the right ones
        }
        if (QM == 1) { h_ca[j] = ca; h_cb[j] = cb; }
      }
    }
    static_assert(NPI <= 64 && NQI <= 64 && NHI <= 64, "item masks are 64 bits");
    // the vector staging path can only be masked by the image's top/bottom border (and by a ragged last tile column); the scalar one by anything
    const bool always_slow = !VEC || (a.Wp % TW != 0) || (p_chan != p_all) || (q_chan != q_all);
    mask_t p_ok = 0, q_ok = 0, h_ok = 0;           // bit j: item j of the CURRENT tile lies inside the image (and its channel is real)
    bool edge = false;                             // the current tile needs masks (wave-uniform)
    int p_base = 0, q_base = 0;                    // element offset of the tile origin inside one image
    auto set_tile = [&](int tile, int& img) {
      const int tx = tile % a.tiles_x, t2 = tile / a.tiles_x;
      const int ty = t2 % a.tiles_y;
      img = t2 / a.tiles_y;
      const int y0 = ty * TH, x0 = tx * TW;
      p_base = y0 * a.Wp + x0;
      q_base = QUPS ? ((y0 >> 1) * a.Wq + (x0 >> 1)) : (y0 * S * a.Wq + x0 * S);
      // logical Q coordinates of the tile origin: (y0*S - PAD, x0*S); an item at (r, col_rel) is inside iff 0 <= origin + offset < size
      const int ylo = PAD - y0 * S, yhi = HqL - y0 * S + PAD;          // ylo <= r < yhi
      const int xlo = 16 - x0 * S, xhi = WqL - x0 * S + 16;            // xlo <= (col_rel + 16) < xhi
      auto inside = [&](int rc) { const int r = rc >> 16, c = rc & 0xFFFF; return (r >= ylo) && (r < yhi) && (c >= xlo) && (c < xhi); };
      edge = always_slow || (y0 + TH > a.Hp) || (ylo > 0) || (yhi < QH);
      if (edge) {
        const int rows_left = a.Hp - y0, cols_left = a.Wp - x0;
        p_ok = 0;
#pragma unroll
        for (int j = 0; j < NPI; ++j)
          p_ok |= (mask_t)(((p_rc[j] >> 16) < rows_left) && ((p_rc[j] & 0xFFFF) < cols_left) ? 1 : 0) << j;
        p_ok &= p_chan;
        q_ok = 0;
#pragma unroll
        for (int j = 0; j < NQI; ++j) q_ok |= (mask_t)(inside(q_rc[j]) ? 1 : 0) << j;
        q_ok &= q_chan;
      } else {
        p_ok = p_chan; q_ok = q_chan;
      }
      if constexpr (NHI > 0) {                     // the halo columns leave the image at its left / right border: always masked
        h_ok = 0;
#pragma unroll
        for (int j = 0; j < NHI; ++j) h_ok |= (mask_t)(inside(h_rc[j]) ? 1 : 0) << j;
        h_ok &= h_chan;
      }
    };
    // Two register sets: the loads of tile p+2 are issued right after tile p has been written to LDS, so a tile's global-memory latency
    // is covered by a full iteration (LDS-store phase + MFMA phase of the consumers).
    struct Regs { float p[NPI][VW], p2[PM == 2 ? NPI : 1][VW], q[NQI][VW], h[NHI > 0 ? NHI : 1]; mask_t p_ok, q_ok, h_ok; bool edge; };
    Regs R[2];
    auto load_tile = [&](int img, Regs& r) {
      const float* pn = a.p + (size_t)img * a.M * p_plane + p_base;
      const float* p2n = (PM == 2) ? a.p2 + (size_t)img * a.M * p_plane + p_base : nullptr;
      const float* qn = a.q + (size_t)img * a.Nq * q_plane + q_base;
      r.p_ok = p_ok; r.q_ok = q_ok; r.h_ok = h_ok; r.edge = edge;
      const bool live = !(a.dbg & 2);
      // masked items load the (valid, aligned) tile origin instead and are zeroed when written to LDS: no divergent branch per item
#pragma unroll
      for (int j = 0; j < NPI; ++j) {
        int off = p_off[j];
        if (edge) off = (((p_ok >> j) & 1u) && live) ? off : 0;
        if constexpr (VEC) {
          const float4 v = *reinterpret_cast<const float4*>(pn + off);
          r.p[j][0] = v.x; r.p[j][1] = v.y; r.p[j][2] = v.z; r.p[j][3] = v.w;
          if constexpr (PM == 2) {
            const float4 u = *reinterpret_cast<const float4*>(p2n + off);
            r.p2[j][0] = u.x; r.p2[j][1] = u.y; r.p2[j][2] = u.z; r.p2[j][3] = u.w;
          }
        } else {
          r.p[j][0] = pn[off];
          if constexpr (PM == 2) r.p2[j][0] = p2n[off];
        }
      }
#pragma unroll
      for (int j = 0; j < NQI; ++j) {
        int off = q_off[j];
        if (edge) off = (((q_ok >> j) & 1u) && live) ? off : 0;
        if constexpr (VEC && QUPS) {
          const float2 v = *reinterpret_cast<const float2*>(qn + off);
          r.q[j][0] = v.x; r.q[j][1] = v.x; r.q[j][2] = v.y; r.q[j][3] = v.y;
        } else if constexpr (VEC) {
          const float4 v = *reinterpret_cast<const float4*>(qn + off);
          r.q[j][0] = v.x; r.q[j][1] = v.y; r.q[j][2] = v.z; r.q[j][3] = v.w;
        } else {
          r.q[j][0] = qn[off];
        }
      }
      if constexpr (NHI > 0) {
#pragma unroll
        for (int j = 0; j < NHI; ++j) r.h[j] = qn[(((h_ok >> j) & 1u) && live) ? h_off[j] : 0];
      }
    };
    const float q_slope = a.slope;
    auto store_tile = [&](float* buf, Regs& r, auto edge_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      float* pl = buf;
      float* ql = buf + 16 * AB * PSP;
#pragma unroll
      for (int j = 0; j < NPI; ++j) {
        const bool full = (j + 1) * 256 <= G::P_ITEMS;               // every lane has an item j (folds after unrolling)
        if (!full && p_lds[j] < 0) continue;
        float v[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          v[e] = r.p[j][e];
          if constexpr (PM == 2) v[e] = p_ca[j] * v[e] + (p_cb[j] * r.p2[j][e] + p_cc[j]);
          if constexpr (EDGE) v[e] = ((r.p_ok >> j) & 1u) ? v[e] : 0.f;          // pixels outside the image contribute nothing
        }
        float* dst = pl + p_lds[j];
        if constexpr (VEC && G::WIDE) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);        // planes and rows are 16-byte aligned in the wide layout
        } else if constexpr (VEC) {
          *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
          *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
        } else {
          dst[0] = v[0];
        }
      }
#pragma unroll
      for (int j = 0; j < NQI; ++j) {
        const bool full = (j + 1) * 256 <= G::Q_ITEMS;
        if (!full && q_lds[j] < 0) continue;
        float v[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          v[e] = r.q[j][e];
          if constexpr (QM == 1) v[e] = leaky(q_ca[j] * v[e] + q_cb[j], q_slope);
          if constexpr (EDGE) v[e] = ((r.q_ok >> j) & 1u) ? v[e] : 0.f;          // zero padding pads the ACTIVATED tensor
        }
        float* dst = ql + q_lds[j];
        if constexpr (VEC && S == 1 && PAD == 1) {     // interior starts at LDS column 1 (column 0 is the left halo): not 8-byte aligned.
          // Four dword stores (the compiler pairs them into ds_write2_b32, which takes any two registers).  A ds_write_b64 of (v1, v2) needs
          // an even-aligned register pair: the compiler then copies the middle of the loaded quad right after the global load - i.e. it waits
          // for the data a full iteration early (seen as s_waitcnt vmcnt(1) straight after the loads).
          dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        } else if constexpr (VEC && S == 1) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
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
          float v = r.h[j];
          if constexpr (QM == 1) v = leaky(h_ca[j] * v + h_cb[j], q_slope);
          ql[h_lds[j]] = ((r.h_ok >> j) & 1u) ? v : 0.f;
        }
      }
    };
    int img;
    set_tile(t_begin, img);
    load_tile(img, R[0]);
    if (T > 1) { set_tile(t_begin + 1, img); load_tile(img, R[1]); }
    lds_barrier();                                    // barrier #0 (matched by the consumers)
#ifdef MS_WGRAD_TRACE_BUILD
    const bool tr = (a.trace != nullptr) && (blockIdx.x == 0) && (threadIdx.x == 256);
#else
    constexpr bool tr = false;      // cycle stamps compiled out: their global stores make the compiler's vmcnt waits conservative
#endif
    // Unrolled by two in the SOURCE: each half owns one register set and one LDS buffer, so no register set is ever selected at run
    // time and the compiler's s_waitcnt vmcnt counts see "tile p's loads are older than tile p+1's" (one barrier per tile, as the consumers).
    auto half = [&](int p, float* buf, Regs& r) {
      if (tr) a.trace[64 + p * 4 + 0] = clock64();
      if (!(a.dbg & 4)) {                                 // tile p, requested one full iteration ago
        if (r.edge) store_tile(buf, r, std::true_type{}); else store_tile(buf, r, std::false_type{});
      }
      if (tr) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a.trace[64 + p * 4 + 1] = clock64(); }
      if (p + 2 < T) { set_tile(t_begin + p + 2, img); load_tile(img, r); }
      if (tr) a.trace[64 + p * 4 + 2] = clock64();
      lds_barrier();                                      // barrier #(p+1): tile p visible; consumers done with tile p-1
      if (tr) a.trace[64 + p * 4 + 3] = clock64();
    };
    for (int p = 0; p < T; p += 2) {
      half(p, smem, R[0]);
      if (p + 1 < T) half(p + 1, smem + BUF, R[1]);
    }
  } else {
    // =========================================== CONSUMER waves ===========================================
    const int mn = lane & 15, k = lane >> 4;
    wg_f32x4 acc[AB][BB][TAPS];
#pragma unroll
    for (int i = 0; i < AB; ++i)
#pragma unroll
      for (int j = 0; j < BB; ++j)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[i][j][t] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
    // ---------------- stride 2: interleaved k index (lane group k multiplies pixel 4j+k in step j), one ds_read_b32 per operand ----------------
    const int a_lane = mn * PSP + wave * TW + k;
    const int b_lane = 16 * AB * PSP + mn * PSQ + wave * S * RSQ + k;
    constexpr int NSTEP = (TW / 16) * 4;
    auto frag_load = [&](const float* ap, const float* bp, int step, float (&af)[AB], float (&bf)[BB][TAPS]) {
      const int x = (step >> 2) * 16 + 4 * (step & 3);
#pragma unroll
      for (int i = 0; i < AB; ++i) af[i] = ap[i * 16 * PSP + x];
#pragma unroll
      for (int jb = 0; jb < BB; ++jb)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) bf[jb][t] = bp[jb * 16 * PSQ + G::tap_off(t) + x];
    };
    // Software pipeline: the fragments of the next step are requested before the MFMAs of the current one are issued, so every LDS round
    // trip has a full MFMA group to land in (with one MFMA wave per SIMD nothing else would cover it).  sched_barrier keeps the compiler
    // from sinking the loads back next to their uses.
    auto compute_narrow = [&](const float* buf) {
      const float* ap = buf + a_lane;
      const float* bp = buf + b_lane;
      float af[2][AB], bf[2][BB][TAPS];
      frag_load(ap, bp, 0, af[0], bf[0]);
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        if (step + 1 < NSTEP) frag_load(ap, bp, step + 1, af[(step + 1) & 1], bf[(step + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
          for (int i = 0; i < AB; ++i)
#pragma unroll
            for (int jb = 0; jb < BB; ++jb)
              acc[i][jb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[step & 1][i], bf[step & 1][jb][t], acc[i][jb][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // ---------------- stride 1: blocked k index (lane group k owns pixels 4k..4k+3 of the segment), 16- and 8-byte LDS reads ----------------
    // a row-step = (16-pixel segment, kernel row ky): 4 k-steps x KS taps x AB x BB MFMAs on one A quad and one 4+KS-1 wide B window per block
    constexpr int NROW = (TW / 16) * KS;
    const int aw_lane = mn * PSP + wave * TW + 4 * k;
    const int bw_lane = 16 * AB * PSP + mn * PSQ + wave * RSQ + 4 * k;
    auto a_load = [&](const float* ap, int seg, float (&af)[AB][4]) {
#pragma unroll
      for (int i = 0; i < AB; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(ap + i * 16 * PSP + seg * 16);
        af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
      }
    };
    auto b_load = [&](const float* bp, int seg, int ky, float (&bf)[BB][4 + KS - 1]) {
#pragma unroll
      for (int jb = 0; jb < BB; ++jb) {
        const float* q = bp + jb * 16 * PSQ + ky * RSQ + seg * 16;
        const float4 v = *reinterpret_cast<const float4*>(q);
        bf[jb][0] = v.x; bf[jb][1] = v.y; bf[jb][2] = v.z; bf[jb][3] = v.w;
        if constexpr (KS == 3) { const float2 w = *reinterpret_cast<const float2*>(q + 4); bf[jb][4] = w.x; bf[jb][5] = w.y; }
      }
    };
    auto compute_wide = [&](const float* buf) {
      const float* ap = buf + aw_lane;
      const float* bp = buf + bw_lane;
      float af[2][AB][4], bf[2][BB][4 + KS - 1];
      a_load(ap, 0, af[0]);
      b_load(bp, 0, 0, bf[0]);
#pragma unroll
      for (int rs = 0; rs < NROW; ++rs) {
        const int seg = rs / KS, ky = rs % KS;
        if (rs + 1 < NROW) {
          const int nseg = (rs + 1) / KS, nky = (rs + 1) % KS;
          b_load(bp, nseg, nky, bf[(rs + 1) & 1]);
          if (nky == 0) a_load(ap, nseg, af[nseg & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int i = 0; i < AB; ++i)
#pragma unroll
              for (int jb = 0; jb < BB; ++jb)
                acc[i][jb][ky * KS + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[seg & 1][i][j], bf[rs & 1][jb][j + kx], acc[i][jb][ky * KS + kx], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto compute = [&](const float* buf) {
      if constexpr (G::WIDE) compute_wide(buf); else compute_narrow(buf);
    };
    lds_barrier();                                      // barrier #0
    lds_barrier();                                      // barrier #1: tile 0 is in buffer 0
#ifdef MS_WGRAD_TRACE_BUILD
    const bool tr = (a.trace != nullptr) && (blockIdx.x == 0) && (threadIdx.x == 0);
#else
    constexpr bool tr = false;
#endif
    for (int p = 0; p < T; ++p) {
      if (tr) a.trace[p * 3 + 0] = clock64();
      if (!(a.dbg & 1)) compute(smem + (p & 1) * BUF);
      if (tr) a.trace[p * 3 + 1] = clock64();
      if (p + 1 < T) lds_barrier();                     // barrier #(p+2)
      if (tr) a.trace[p * 3 + 2] = clock64();
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

template <int KS, int S, int AB, int BB, int TW, bool VEC, bool QUPS, int PM, int QM>
int launch_wgrad(WgArgs a, int max_wg, hipStream_t st) {
  using G = WgGeo<KS, S, AB, BB, TW, VEC, QUPS>;
  const size_t lds_bytes = sizeof(float) * (size_t)G::LDS_FLOATS;
  static_assert(sizeof(float) * (size_t)G::LDS_FLOATS <= 160 * 1024, "tile does not fit the LDS");
  static std::once_flag attr_once;                     // one flag per instantiation (no unsynchronised mutable state in the ABI)
  std::call_once(attr_once, []() { (void)hipFuncSetAttribute((const void*)wgrad_mfma_kernel<KS, S, AB, BB, TW, VEC, QUPS, PM, QM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)); });
  a.tiles_x = cdiv(a.Wp, TW); a.tiles_y = cdiv(a.Hp, G::TH);
  a.ntiles = a.N * a.tiles_x * a.tiles_y;
  const int npm = cdiv(a.M, 16 * AB);
  a.npairs_n = cdiv(a.Nq, 16 * BB);
  const int npairs = npm * a.npairs_n;
  int nslots = std::max(1, std::min(a.ntiles, max_wg / npairs));
  a.per = cdiv(a.ntiles, nslots);
  nslots = cdiv(a.ntiles, a.per);                      // no empty workgroup
  a.nslots = nslots;
  MS_LAUNCH((wgrad_mfma_kernel<KS, S, AB, BB, TW, VEC, QUPS, PM, QM>), dim3((unsigned)(npairs * nslots)), dim3(512), lds_bytes, st, a);
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
