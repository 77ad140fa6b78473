// Second generation of the 3x3 stride-2 forward convolution (res_convdown.down, encoder_decoder.py:40) for gfx950 - the first generation is
// conv_mfma_kernel<3, 2, ...> (ms_conv_kernel.h): staging waves that de-interleave even / odd columns through registers, 8 x 32-pixel output tiles (52 % / 62 % full
// on 20 x 20 / 40 x 40 outputs), 4-channel chunks: 0.28-0.5 of the MFMA bound at config 4 (profiles/r04_experiments.txt 11).
//
// Same recipe as the sub-pixel kernel's second generation (ms_conv_subpix2.h): the MFMAs and their order per output element are the first generation's (tap-major inside a
// 4-channel group, groups ascending: same bits), everything around them changed:
//   * nothing is staged through registers: the input patch and the 9-tap weight slice of a 4-channel chunk travel HBM/L2 -> LDS by LDS-DMA in 16-byte pieces (zeros from the
//     buffer bounds check), the patch keeps its columns INTERLEAVED - an A fragment is one ds_read_b64 at an even column (taps kx = 1, 2 of 16 output pixels) + one ds_read_b32
//     (tap kx = 0) per kernel row instead of three stride-2 32-bit reads;
//   * NT = 1, 2 or 4 sixteen-channel output blocks per staged patch (the deep layers re-read their patch from L2 once per 64 output channels, not once per 32);
//   * GEO 0: 8 x 32-pixel output tiles.  GEO 1: sixteen independent 4 x 4-pixel output blocks per work item from a flattened (image, block row, block column) list, each
//     with its own 9 x 12 patch: every output size that is a multiple of 4 fills its MFMA rows (20 x 20: 52 % -> 100 %).
// fp32 storage, bias, no statistics.  PRO 0: no prologue (the block-input convs of down2 .. down4).  PRO 1 (round 5; GEO 0 only): v = lrelu(a[c] v + b[c]) per input channel
// - down1's input is the never-materialised `inc` activation (37.8 us at config 2 on the first generation, profiles/r05_step_budget_c2.txt) - applied by the MFMA waves to
// the fragments they read (the patch still travels by DMA; a staged value is activated once per output pixel that uses it, 2.25 x on average: 3 vector instructions per
// element next to the MFMA it feeds), the coefficients from a table in LDS (given, or derived in the launch: `_xfin` kind 0); zero padding pads the ACTIVATED tensor:
// with an even input only kernel row 0 of output row 0 and kernel column 0 of output column 0 leave the image - one wave-uniform and one per-lane select.
#pragma once
#include <cstdlib>
#include <map>
#include <mutex>
#include "ms_conv_kernel.h"

namespace ms {

template <int GEO, int NT>
struct S2Geo {
  static constexpr int CK = 4;
  static constexpr int WSW = (NT == 1) ? 16 : 16 * NT + 16;             // weight row stride (floats): B fragments of the four k-rows on disjoint banks
  static constexpr int WROWP = WSW / 4;                                 // 16-byte pieces per weight row (NT > 1: the last four are padding)
  static constexpr int W_ITEMS = 9 * CK * WROWP;
  static constexpr int NWJ = ((W_ITEMS + 63) / 64 + 3) / 4;
  static constexpr int W_FLOATS = 4 * NWJ * 256;
  // GEO 0: patch of an 8 x 32 output tile: input rows 2y0-1 .. 2y0+15 (17), columns 2x0-4 .. 2x0+67 (18 pieces); channel plane padded to 312 pieces = 1248 floats
  // (== 32 mod 64: the two k-rows a 64-bit read handles per pass hit disjoint banks)
  // GEO 1: patch of a 4 x 4 output block: 9 rows x 12 columns (3 pieces), padded to 28 pieces = 112 floats per channel
  static constexpr int ROWP = (GEO == 0) ? 18 : 3, ROWS = (GEO == 0) ? 17 : 9;
  static constexpr int RS = ROWP * 4;
  static constexpr int PLANEP = (GEO == 0) ? 312 : 28;
  static constexpr int PS = PLANEP * 4;
  static constexpr int BSB = CK * PS;                                   // GEO 1: floats per block
  static constexpr int IN_ITEMS = (GEO == 0) ? CK * PLANEP : 16 * CK * PLANEP;
  static constexpr int NJ = ((IN_ITEMS + 63) / 64 + 3) / 4;             // 5 | 7 DMA instructions per staging wave and chunk
  static constexpr int IN_FLOATS = 4 * NJ * 256;
  static constexpr int BUF = IN_FLOATS + W_FLOATS;
  static constexpr int KDMA = NJ + NWJ;
  static constexpr int NBUF = (GEO == 0) ? 3 : 2;
  static constexpr int OOB = (int)0x80000000;
};

template <int GEO, int NT, int PRO>
__global__ __launch_bounds__(512, 4) void conv_s2_kernel(const ConvArgs a) {
  static_assert(PRO == 0 || GEO == 0, "the prologue form is built for the tile geometry");
  using G = S2Geo<GEO, NT>;
  constexpr int CK = G::CK, WSW = G::WSW, RS = G::RS, PS = G::PS, BSB = G::BSB, BUF = G::BUF, NJ = G::NJ, NWJ = G::NWJ, OOB = G::OOB, NBUF = G::NBUF;
  constexpr int COUT_TILE = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = MS_TID >> 6, lane = MS_TID & 63;
  const bool producer = wave >= 4;
  const int ncb = a.ncb;
  const int ntiles = a.tiles_x * a.tiles_y;                     // GEO 1: groups of 16 blocks over the whole batch (tiles_y = 1)
  const int nitems = (GEO == 0 ? a.N : 1) * ntiles * ncb;
  const int nchunks = a.cin_pad / CK;
  const int nbx = a.Wout >> 2, nby = (a.Hout + 3) >> 2, NB = a.N * nbx * nby;
  const int vb = ((int)gridDim.x % 8 == 0) ? (((int)blockIdx.x % 8) * ((int)gridDim.x / 8) + (int)blockIdx.x / 8) : (int)blockIdx.x;
  const int my_items = (nitems - vb + (int)gridDim.x - 1) / (int)gridDim.x;
  const int T = my_items * nchunks;
  const int plane = a.Hs * a.Ws;
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto decode = [&](int it, int& n, int& tile, int& cb) { cb = it % ncb; const int t2 = it / ncb; tile = t2 % ntiles; n = t2 / ntiles; };
  float* cf_lds = smem + NBUF * BUF;                     // PRO 1: [cin_pad][4] prologue coefficients

  if (PRO != 0 && a.xf_tab == nullptr) {
    for (int c = MS_TID; c < a.cin_pad; c += 512) {
      float ca = 1.f, cb_ = 0.f;
      if (c < a.Cin) { ca = a.pro_a[c * a.pro_cstride]; cb_ = a.pro_b[c * a.pro_cstride]; }
      reinterpret_cast<float4*>(cf_lds)[c] = make_float4(ca, cb_, 0.f, 0.f);
    }
  }

  if (producer) {
    // =========================================== STAGING waves: LDS-DMA only ===========================================
    __builtin_amdgcn_s_setprio(3);
    const int sw = __builtin_amdgcn_readfirstlane(wave) - 4;
    const ms_i32x4 rs_in = ms_dma_rsrc_n(a.in, (unsigned)a.N * a.Cin * plane * 4u);
    const ms_i32x4 rs_w = ms_dma_rsrc_n(a.w, 9u * a.cin_pad * a.cout_pad * 4u);
    const unsigned lds0 = ms_lds_addr(smem);
    // weight slice: piece -> (tap q, channel c, 4 output channels); LDS row (q * CK + c) of WSW floats
    int w_voff[NWJ];
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
      const int idx = (sw + 4 * j) * 64 + lane;
      const int row = idx / G::WROWP, p = idx - row * G::WROWP, c = row % CK, q = row / CK;
      w_voff[j] = (idx < G::W_ITEMS && p < 4 * NT) ? (int)((((size_t)q * a.cin_pad + c) * a.cout_pad + 4 * p) * 4) : OOB;
    }
    int i_pk[NJ], i_voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int idx = (sw + 4 * j) * 64 + lane;
      if (GEO == 0) {
        const int c = idx / G::PLANEP, rem = idx - c * G::PLANEP, r = rem / G::ROWP, f = rem - r * G::ROWP;
        i_pk[j] = (idx < G::IN_ITEMS && rem < G::ROWS * G::ROWP) ? ((c << 16) | (r << 8) | f) : -1;
      } else {
        const int blk = idx / (CK * G::PLANEP), rem = idx - blk * (CK * G::PLANEP), c = rem / G::PLANEP, p = rem - c * G::PLANEP, r = p / 3, f = p - r * 3;
        i_pk[j] = (idx < G::IN_ITEMS && p < 27) ? ((blk << 16) | (c << 8) | (r << 4) | f) : -1;
      }
      i_voff[j] = OOB;
    }
    auto set_tile0 = [&](int tile) {
      const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
      const int Y0 = 2 * ty * 8 - 1, X0 = 2 * tx * 32 - 4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int pk = i_pk[j];
        const int Y = Y0 + ((pk >> 8) & 0xFF), X = X0 + 4 * (pk & 0xFF);
        const bool ok = (pk >= 0) && (Y >= 0) && (Y < a.Hs) && (X >= 0) && (X < a.Ws);          // Ws % 4 == 0: a piece is inside or outside as a whole
        i_voff[j] = ok ? (((pk >> 16) * plane + Y * a.Ws + X) * 4) : OOB;
      }
    };
    auto set_group1 = [&](int grp) {
      const int b = grp * 16 + (lane & 15);
      const int bx = b % nbx, t = b / nbx, by = t % nby, n = t / nby;
      const int Y0 = 8 * by - 1, X0 = 8 * bx - 4;
      const int base = (b < NB) ? ((n * a.Cin) * plane + Y0 * a.Ws + X0) : OOB;
      const int yx = (b < NB) ? (((Y0 + 64) << 16) | (X0 + 64)) : 0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int pk = i_pk[j];
        const int blk = (pk >> 16) & 0xF;
        const int bb = __shfl(base, blk, 64), byx = __shfl(yx, blk, 64);
        const int r = (pk >> 4) & 0xF, f = pk & 0xF, c = (pk >> 8) & 0xFF;
        const int Y = (byx >> 16) - 64 + r, X = (byx & 0xFFFF) - 64 + 4 * f;
        const bool ok = (pk >= 0) && (bb != OOB) && (Y >= 0) && (Y < a.Hs) && (X >= 0) && (X < a.Ws);
        i_voff[j] = ok ? ((bb + c * plane + r * a.Ws + 4 * f) * 4) : OOB;
      }
    };
    auto issue = [&](int buf, int n, int cb, int chunk) {
      const unsigned lb = lds0 + (unsigned)buf * (BUF * 4);
      const int c0 = chunk * CK;
      const int soff_w = (c0 * a.cout_pad + cb * COUT_TILE) * 4;
#pragma unroll
      for (int j = 0; j < NWJ; ++j) ms_lds_dma16(rs_w, lb + G::IN_FLOATS * 4 + (unsigned)(sw + 4 * j) * 1024, w_voff[j], soff_w);
      const int soff_i = ((GEO == 0 ? n * a.Cin : 0) + c0) * plane * 4;
      const bool ctail = (c0 + CK > a.Cin);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        int v = i_voff[j];
        if (ctail) { const int c = (GEO == 0) ? (i_pk[j] >> 16) : ((i_pk[j] >> 8) & 0xFF); if (c0 + c >= a.Cin) v = OOB; }
        ms_lds_dma16(rs_in, lb + (unsigned)(sw + 4 * j) * 1024, v, soff_i);
      }
    };
    int item = vb, chunk = 0, n, tile, cb, tile_set = -1, ring = 0;
    decode(item, n, tile, cb);
    auto issue_next = [&](bool more) {
      if (tile != tile_set) { if (GEO == 0) set_tile0(tile); else set_group1(tile); tile_set = tile; }
      issue(ring, n, cb, chunk);
      if (++ring == NBUF) ring = 0;
      if (++chunk == nchunks) { chunk = 0; item += gridDim.x; if (more) decode(item, n, tile, cb); }
    };
    lds_barrier();                                    // barrier #0
    if (NBUF == 3) issue_next(T > 1);
    for (int p = 0; p < T; ++p) {
      if (NBUF == 3) {
        if (p + 1 < T) { issue_next(p + 2 < T); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::KDMA) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        issue_next(p + 1 < T);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      lds_barrier();                                  // barrier #(p+1)
    }
    return;
  }

  // =========================================== MFMA waves ===========================================
  if (PRO != 0 && a.xf_tab != nullptr) {
    // the coefficients feed this launch's PROLOGUE: in LDS before the first chunk is read (barrier #0 / #1)
    unsigned xf_tag = 0u;
    int xf_nparts = 0;
    xfin_header(a, xf_tag, xf_nparts);
    if (!xfin_produce(a, xf_tag, xf_nparts)) __builtin_amdgcn_s_sleep(30);
    xfin_fill<2>(a, cf_lds, a.cin_pad, xf_tag, vb & (kXfinRep - 1), MS_TID, 256, 1.f, 0.f);
  }
  const int m = lane & 15, k = lane >> 4;
  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // LDS float index of the lane's pixel of M-tile 0, kernel row 0, patch column of tap kx = 1 (even: the 64-bit read takes kx = 1, 2; kx = 0 is the float in front)
  //   GEO 0: M-tile i = output row 2*wave + (i >> 1), columns (i & 1)*16 + m;  GEO 1: M-tile i = block 4*wave + i, pixel (m >> 2, m & 3)
  const int a_lane = (GEO == 0) ? (k * PS + (2 * (2 * wave)) * RS + 2 * m + 4) : (4 * wave * BSB + k * PS + 2 * (m >> 2) * RS + 2 * (m & 3) + 4);
  auto mt_off = [](int i) { return (GEO == 0) ? (2 * (i >> 1) * RS + (i & 1) * 32) : (i * BSB); };
  const int b_lane = G::IN_FLOATS + k * WSW + m;

  bool pad_top = false, pad_left = false;                // PRO 1: this tile holds output row 0 / output column 0
  auto set_tile = [&](int tile) { pad_top = (tile / a.tiles_x == 0) && (wave == 0); pad_left = (tile % a.tiles_x == 0) && (m == 0); };
  auto compute = [&](const float* buf, int chunk) {
    float bf[9][NT];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[q][j] = buf[b_lane + q * CK * WSW + 16 * j];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float af[3][3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float2 v = *reinterpret_cast<const float2*>(buf + a_lane + mt_off(i) + ky * RS);
        af[ky][0] = buf[a_lane + mt_off(i) + ky * RS - 1]; af[ky][1] = v.x; af[ky][2] = v.y;
      }
      if constexpr (PRO == 1) {
        const float4 cf = reinterpret_cast<const float4*>(cf_lds)[chunk * CK + k];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            float v = leaky(cf.x * af[ky][kx] + cf.y, a.slope);
            if (ky == 0 && (i >> 1) == 0) v = pad_top ? 0.f : v;          // input row -1
            if (kx == 0 && (i & 1) == 0) v = pad_left ? 0.f : v;         // input column -1
            af[ky][kx] = v;
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      // first generation's order: taps ascending (ky outer, kx inner)
#pragma unroll
      for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q / 3][q % 3], bf[q][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  float bias_v[NT];
  int bias_cb = -1;
  auto load_bias = [&](int cb) {
    if (cb == bias_cb) return;
    bias_cb = cb;
#pragma unroll
    for (int j = 0; j < NT; ++j) { const int co = cb * COUT_TILE + j * 16 + m; bias_v[j] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f; }
  };
  // D layout: this lane holds output channel m of the M-tile's pixels 4k .. 4k+3 - four consecutive columns in both geometries
  auto epilogue = [&](int n_item, int tile, int cb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int n, y, x;
      bool ok;
      if (GEO == 0) {
        const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
        n = n_item; y = ty * 8 + 2 * wave + (i >> 1); x = tx * 32 + (i & 1) * 16 + 4 * k;
        ok = y < a.Hout && x < a.Wout;
      } else {
        const int b = tile * 16 + 4 * wave + i;
        const int bx = b % nbx, t = b / nbx, by = t % nby;
        n = t / nby; y = by * 4 + k; x = bx * 4;
        ok = b < NB && y < a.Hout;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = cb * COUT_TILE + j * 16 + m;
        if (ok && co < a.Cout) {
          const size_t off = (((size_t)n * a.Cout + co) * a.Hout + y) * a.Wout + x;
          *reinterpret_cast<float4*>(a.out + off) = make_float4(acc[i][j][0] + bias_v[j], acc[i][j][1] + bias_v[j], acc[i][j][2] + bias_v[j], acc[i][j][3] + bias_v[j]);
        }
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  int item = vb, chunk = 0, n, tile, cb, ring = 0;
  decode(item, n, tile, cb);
  load_bias(cb);
  set_tile(tile);
  lds_barrier();                                      // barrier #0
  lds_barrier();                                      // barrier #1: chunk 0 is in buffer 0
  for (int p = 0; p < T; ++p) {
    compute(smem + ring * BUF, chunk);
    if (++ring == NBUF) ring = 0;
    if (chunk + 1 == nchunks) {
      epilogue(n, tile, cb);
      chunk = 0; item += gridDim.x;
      if (p + 1 < T) { decode(item, n, tile, cb); load_bias(cb); set_tile(tile); }
    } else {
      ++chunk;
    }
    if (p + 1 < T) lds_barrier();
  }
}

int conv_s2g2_switch();      // ms_conv.hip: option "conv.s2g2" (0: off - A/B runs and the same-bits tests)
inline void conv_s2g2_plan(const ConvArgs& a, int& geo, int& nt);
inline bool conv_s2g2_eligible(const ConvArgs& a, int ks, int stride, int fetch) {
  if (conv_s2g2_switch() == 0 || ks != 3 || stride != 2 || fetch != FETCH_NORMAL || a.act_bf16 != 0 || a.pro_mode > 1 || a.in2 != nullptr || a.stats != nullptr ||
      a.epi_mode != 0 || a.ride_out != nullptr) return false;
  if (a.pro_mode == 0 && a.xf_tab != nullptr) return false;
  if (a.pro_mode == 1) {                     // per-channel coefficients, the tile geometry, a BatchNorm-apply record when derived in the launch
    int geo, nt;
    conv_s2g2_plan(a, geo, nt);
    if (geo != 0 || a.pro_nstride != 0 || (a.xf_tab != nullptr && a.xf_kind != 0) || a.Hs != 2 * a.Hout || a.Ws != 2 * a.Wout) return false;
  }
  if (a.Ws % 8 != 0 || a.Hs % 2 != 0 || a.Cin % 4 != 0 || a.Cin < 16) return false;                    // 16-byte pieces on both sides, whole 4-channel chunks
  if (a.Wout <= 16) return false;            // the 16-pixel level keeps the first generation's 4 x 16 tiles (128 -> 128 @16x32x32 -> 16x16: 30 against 33 us; tools/ab_subpix.py)
  if ((long long)a.N * a.Cin * a.Hs * a.Ws * 4 >= (1LL << 31) || 9LL * a.cin_pad * a.cout_pad * 4 >= (1LL << 31)) return false;
  return aligned16(a.in) && aligned16(a.out) && aligned16(a.w);
}

// geometry and channel blocks per patch: blocks where the 8 x 32 tiles are poorly filled and the block list has work for every CU; as many channel blocks as leave
// >= 1.5 work items per CU (the patch is re-read from L2 once per item)
inline void conv_s2g2_plan(const ConvArgs& a, int& geo, int& nt) {
  const double fill_t = (double)a.Hout * a.Wout / ((double)cdiv(a.Hout, 8) * 8 * cdiv(a.Wout, 32) * 32);
  const double fill_b = (double)a.Hout / (cdiv(a.Hout, 4) * 4);
  const long groups_b = cdiv((long)a.N * (a.Wout / 4) * cdiv(a.Hout, 4), 16L), tiles_t = (long)a.N * cdiv(a.Hout, 8) * cdiv(a.Wout, 32);
  geo = (a.Wout % 4 == 0 && fill_b >= 1.3 * fill_t && groups_b * cdiv(a.Cout, 16) >= num_cus()) ? 1 : 0;
  const long units = geo ? groups_b : tiles_t;
  nt = 1;
  for (int cand : {4, 2}) if (a.Cout >= 16 * cand && 2 * units * cdiv(a.Cout, 16 * cand) >= 3L * num_cus()) { nt = cand; break; }      // (>= 1.5 work items per CU)
}

template <int GEO, int NT, int PRO>
int launch_conv_s2_t(ConvArgs a, hipStream_t st) {
  using G = S2Geo<GEO, NT>;
  const size_t lds_bytes = sizeof(float) * (G::NBUF * (size_t)G::BUF + (PRO ? 4 * (size_t)a.cin_pad : 0));
  static std::once_flag attr_once;
  std::call_once(attr_once, []() { (void)hipFuncSetAttribute((const void*)conv_s2_kernel<GEO, NT, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)); });
  a.ncb = cdiv(a.Cout, 16 * NT);
  long nitems;
  if (GEO == 0) { a.tiles_x = cdiv(a.Wout, 32); a.tiles_y = cdiv(a.Hout, 8); nitems = (long)a.N * a.tiles_x * a.tiles_y * a.ncb; }
  else { const long nb = (long)a.N * (a.Wout / 4) * cdiv(a.Hout, 4); a.tiles_x = (int)cdiv(nb, 16L); a.tiles_y = 1; nitems = (long)a.tiles_x * a.ncb; }
  const int per_cu = std::max(1, std::min(2, (int)((160 * 1024) / (lds_bytes + 256))));
  long nblocks = std::min<long>(nitems, (long)num_cus() * per_cu);
  if (nblocks > a.ncb) nblocks -= nblocks % a.ncb;
  MS_LAUNCH((conv_s2_kernel<GEO, NT, PRO>), dim3((unsigned)nblocks), dim3(512), lds_bytes, st, a);
  return check_launch("conv_s2");
}

int conv_dispatch_s2g2(const ConvArgs& a, hipStream_t st);      // ms_conv_inst_s2g2.hip

}  // namespace ms
