// Second generation of the sub-pixel convolution kernel (ms_conv_subpix.h has the arithmetic, the (parity, tap) enumeration and the first-generation kernel).
//
// Same products in the same order per output element (so the output tensor has the first generation's bits); what changed is everything around the MFMAs
// (round 4; counters of the first generation at config 4, profiles/r04_experiments.txt 7: 3.3 vector instructions per MFMA, most of them the staging waves
// re-summing the 3x3 taps into the 2x2 sub-pixel weights for every chunk of every work item; 52 % tile fill on 20 x 20 images, 62 % on 40 x 40):
//   * NOTHING is staged through registers: the input patch and the weight slice of a chunk go HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds, ms_lds_dma16 / ms_lds_dma4),
//     out-of-image and out-of-tensor elements come back as zeros from the buffer bounds check (no per-element tests);
//   * MODE 0 takes its 16 sub-pixel weight matrices from an APPENDIX packed once per weight version (ms_subpix_pack: [16][cin_pad][cout_pad], the sums formed in the
//     first generation's order: same bits); MODE 1's nine matrices are taps of the data-gradient layout as they are;
//   * GEO 0: the first generation's 8 x 32-pixel tile (large images).  GEO 1: sixteen independent 4 x 4-pixel blocks per work item, four per MFMA wave, taken from a
//     flattened (image, block row, block column) list, each staged with its own 6 x 6 halo patch: every stored size that is a multiple of 4 fills its M-tiles completely
//     (20 x 20: 52 % -> 100 %, 40 x 40: 62 % -> 100 %) for 2.25x the patch bytes (L2 hits, no vector instruction).
// fp32 activation storage only (bf16 storage needs a conversion on the way to LDS: the first generation keeps those calls).
#pragma once
#include "ms_conv_subpix.h"

namespace ms {

template <int MODE, int GEO>
struct SubGeo2 {
  static constexpr int CK = 8, WS = 16;
  static constexpr int NCOMBO = (MODE == 0) ? 16 : 9;
  static constexpr int TLH = 8, TLW = 32, IH = TLH + 2;
  static constexpr int RS = (GEO == 0) ? 40 : 6;                       // patch row stride (floats)
  static constexpr int PS = (GEO == 0) ? IH * 40 : 36;                 // channel-plane stride
  static constexpr int BS = CK * 36;                                   // GEO 1: block stride (8 channel planes of 6 x 6)
  static constexpr int IN_ITEMS = (GEO == 0) ? CK * IH * 10 : 16 * BS; // DMA lanes per chunk: 16-byte pieces (GEO 0) | single floats (GEO 1)
  static constexpr int NJ = ((IN_ITEMS + 63) / 64 + 3) / 4;            // DMA instructions per staging wave and chunk: 4 | 18 (every wave issues the same number:
  static constexpr int IN_INSTR = 4 * NJ;                              //  the lanes behind IN_ITEMS read zeros into padding) - 16 | 72 wave-level instructions per chunk
  static constexpr int IN_FLOATS = IN_INSTR * ((GEO == 0) ? 256 : 64);
  static constexpr int W_ITEMS = NCOMBO * CK * 4;                      // 16-byte pieces of the weight slice [q][c][16]
  static constexpr int NWJ = ((W_ITEMS + 63) / 64 + 3) / 4;            // 2
  static constexpr int W_INSTR = 4 * NWJ;
  static constexpr int BUF = IN_FLOATS + W_INSTR * 256;
  static constexpr int KDMA = NJ + NWJ;                                // DMA instructions a staging wave has in flight per chunk
  static constexpr int OOB = (int)0x80000000;                          // a byte offset behind every tensor: the buffer unit returns 0
};

template <int MODE, int GEO, int NBUF>
__global__ __launch_bounds__(512, 4) void conv_subpix2_kernel(const ConvArgs a, const float* __restrict__ mk_ref) {
  using G = SubGeo2<MODE, GEO>;
  constexpr int TLH = G::TLH, TLW = G::TLW, CK = G::CK, IH = G::IH, RS = G::RS, PS = G::PS, BS = G::BS, NCOMBO = G::NCOMBO, WS = G::WS, BUF = G::BUF;
  constexpr int NJ = G::NJ, NWJ = G::NWJ, OOB = G::OOB;
  constexpr int NCOMBO_TAPS = (MODE == 0) ? 16 : 9;            // matrices in the weight tensor the DMA reads (appendix | data-gradient taps)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = MS_TID >> 6, lane = MS_TID & 63;
  const bool producer = wave >= 4;
  const int ncb = a.ncb;
  // GEO 0: a.tiles_x x a.tiles_y tiles per image; GEO 1: a.tiles_x = groups of 16 blocks over the whole batch, a.tiles_y = 1, blocks per row / per image column below
  const int ntiles = a.tiles_x * a.tiles_y;
  const int nitems = (GEO == 0 ? a.N : 1) * ntiles * ncb;
  const int nchunks = (a.cin_pad + CK - 1) / CK;
  const int nbx = (a.Ws + 3) >> 2, nby = (a.Hs + 3) >> 2, NB = a.N * nbx * nby;       // (GEO 1; an even width that is not a multiple of 4 ends in a half block)
  const int vb = ((int)gridDim.x % 8 == 0) ? (((int)blockIdx.x % 8) * ((int)gridDim.x / 8) + (int)blockIdx.x / 8) : (int)blockIdx.x;
  const int my_items = (nitems - vb + (int)gridDim.x - 1) / (int)gridDim.x;
  const int T = my_items * nchunks;
  const int plane = a.Hs * a.Ws;
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto decode = [&](int it, int& n, int& tile, int& cb) { cb = it % ncb; const int t2 = it / ncb; tile = t2 % ntiles; n = t2 / ntiles; };

  if (producer) {
    // =========================================== STAGING waves: LDS-DMA only ===========================================
    __builtin_amdgcn_s_setprio(3);
    const int sw = __builtin_amdgcn_readfirstlane(wave) - 4;
    const ms_i32x4 rs_in = ms_dma_rsrc_n(a.in, (unsigned)a.N * a.Cin * plane * 4u);
    const ms_i32x4 rs_w = ms_dma_rsrc_n(MODE == 0 ? a.wu : a.w, (unsigned)NCOMBO_TAPS * a.cin_pad * a.cout_pad * 4u);
    const unsigned lds0 = ms_lds_addr(smem);
    // weight slice: lane -> (product q, channel c, 4 output channels j4); global element ((tap(q) * cin_pad + c0 + c) * cout_pad + co0 + 4 j4)
    int w_voff[NWJ];
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
      const int idx = (sw + 4 * j) * 64 + lane;
      const int j4 = idx & 3, row = idx >> 2, c = row % CK, q = row / CK;
      int tap = q;
      if (MODE == 1) {
#pragma unroll
        for (int qq = 0; qq < NCOMBO; ++qq) if (qq == q) { const SubCombo cq = sub_combo<MODE>(qq); tap = cq.ky0 * 3 + cq.kx0; }
      }
      w_voff[j] = (idx < G::W_ITEMS) ? (int)((((size_t)tap * a.cin_pad + c) * a.cout_pad + 4 * j4) * 4) : OOB;
    }
    // input patch: lane -> element of the chunk's LDS image (linear in the DMA lane index)
    int i_pk[NJ], i_voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int idx = (sw + 4 * j) * 64 + lane;
      if (GEO == 0) {
        const int f = idx % 10, row = idx / 10, r = row % IH, c = row / IH;       // 16-byte piece f of patch row r of channel c
        i_pk[j] = (idx < G::IN_ITEMS) ? ((c << 16) | (r << 8) | f) : -1;
      } else {
        const int blk = idx / BS, rem = idx % BS, c = rem / 36, r = (rem % 36) / 6, cc = rem % 6;
        i_pk[j] = (blk << 16) | (c << 8) | (r << 4) | cc;
      }
      i_voff[j] = OOB;
    }
    const bool ctail = (a.Cin & (CK - 1)) != 0;
    auto set_tile0 = [&](int tile) {
      const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
      const int y0 = ty * TLH - 1, x0 = tx * TLW - 4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int pk = i_pk[j];
        const int Y = y0 + ((pk >> 8) & 0xFF), X = x0 + 4 * (pk & 0xFF);
        const bool ok = (pk >= 0) && (Y >= 0) && (Y < a.Hs) && (X >= 0) && (X < a.Ws);          // Ws % 4 == 0: a piece is inside or outside as a whole
        i_voff[j] = ok ? (((pk >> 16) * plane + Y * a.Ws + X) * 4) : OOB;
      }
    };
    auto set_group1 = [&](int grp) {
      // lanes 0..15 work out their block's image / origin once, everybody picks its blocks' up with a cross-lane read
      const int b = grp * 16 + (lane & 15);
      const int bx = b % nbx, t = b / nbx, by = t % nby, n = t / nby;
      const int base = (b < NB) ? ((n * a.Cin) * plane + (by * 4 - 1) * a.Ws + (bx * 4 - 1)) : OOB;
      const int yx = (b < NB) ? (((by * 4 - 1 + 64) << 16) | (bx * 4 - 1 + 64)) : 0;              // (+64: the fields stay non-negative)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int pk = i_pk[j];
        const int blk = pk >> 16;
        const int bb = __shfl(base, blk, 64), byx = __shfl(yx, blk, 64);
        const int r = (pk >> 4) & 0xF, cc = pk & 0xF, c = (pk >> 8) & 0xFF;
        const int Y = (byx >> 16) - 64 + r, X = (byx & 0xFFFF) - 64 + cc;
        const bool ok = (bb != OOB) && (Y >= 0) && (Y < a.Hs) && (X >= 0) && (X < a.Ws);
        i_voff[j] = ok ? ((bb + c * plane + r * a.Ws + cc) * 4) : OOB;
      }
    };
    auto issue = [&](int buf, int n, int cb, int chunk) {
      const unsigned lb = lds0 + (unsigned)buf * (BUF * 4);
      const int c0 = chunk * CK;
      const int soff_w = (c0 * a.cout_pad + cb * 16) * 4;
#pragma unroll
      for (int j = 0; j < NWJ; ++j)
        ms_lds_dma16(rs_w, lb + G::IN_FLOATS * 4 + (unsigned)(sw + 4 * j) * 1024, w_voff[j], soff_w);
      const int soff_i = ((GEO == 0 ? n * a.Cin : 0) + c0) * plane * 4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        int v = i_voff[j];
        if (ctail) { const int c = (GEO == 0) ? (i_pk[j] >> 16) : ((i_pk[j] >> 8) & 0xFF); if (c0 + c >= a.Cin) v = OOB; }
        if (GEO == 0) ms_lds_dma16(rs_in, lb + (unsigned)(sw + 4 * j) * 1024, v, soff_i);
        else ms_lds_dma4(rs_in, lb + (unsigned)(sw + 4 * j) * 256, v, soff_i);
      }
    };
    int item = vb, chunk = 0, n, tile, cb, tile_set = -1, ring = 0;
    decode(item, n, tile, cb);
    auto issue_next = [&](bool more) {                // the next chunk of this workgroup's sequence -> the next buffer of the ring
      if (tile != tile_set) { if (GEO == 0) set_tile0(tile); else set_group1(tile); tile_set = tile; }
      issue(ring, n, cb, chunk);
      if (++ring == NBUF) ring = 0;
      if (++chunk == nchunks) { chunk = 0; item += gridDim.x; if (more) decode(item, n, tile, cb); }
    };
    lds_barrier();                                    // barrier #0
    if (NBUF == 3) issue_next(T > 1);
    for (int p = 0; p < T; ++p) {
      if (NBUF == 3) {
        // three buffers: chunk p + 1 -> buffer (p + 1) % 3, whose readers (chunk p - 2) left at barrier #p; chunk p (issued one iteration ago) must have landed
        if (p + 1 < T) { issue_next(p + 2 < T); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::KDMA) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        // two buffers: chunk p -> buffer p & 1: the MFMA waves left it (chunk p - 2) at barrier #p, they are on chunk p - 1 while these pieces fly
        issue_next(p + 1 < T);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      lds_barrier();                                  // barrier #(p+1)
    }
    return;
  }

  // =========================================== MFMA waves ===========================================
  const int m = lane & 15, k = lane >> 4;
  f32x4 acc[4][4];                                      // [M-tile i][parity py*2+px]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  // GEO 0: M-tile i = stored row 2*wave + (i >> 1), columns (i & 1)*16 .. +15 (lane m = column).  GEO 1: M-tile i = block 4*wave + i, lane m = pixel (m >> 2, m & 3).
  const int a_lane = (GEO == 0) ? (k * PS + (2 * wave + 1) * RS + 4 + m) : (4 * wave * BS + k * PS + ((m >> 2) + 1) * RS + (m & 3) + 1);
  const int b_lane = G::IN_FLOATS + k * WS + m;
  auto mt_off = [](int i) { return (GEO == 0) ? ((i >> 1) * RS + (i & 1) * 16) : (i * BS); };

  auto load_a = [&](const float* buf, int cg, int i, float (&af)[9]) {
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      if (sub_shift_used<MODE>(s / 3 - 1, s % 3 - 1)) af[s] = buf[a_lane + cg * 4 * PS + mt_off(i) + (s / 3 - 1) * RS + (s % 3 - 1)];
    }
  };
  auto compute = [&](const float* buf, int ncg) {
#pragma unroll
    for (int cg = 0; cg < CK / 4; ++cg) {
      if (cg < ncg) {
        constexpr int NAB = (MODE == 1) ? 2 : 1;
        float bf[NCOMBO], af[NAB][9];
#pragma unroll
        for (int q = 0; q < NCOMBO; ++q) bf[q] = buf[b_lane + (q * CK + cg * 4) * WS];
        load_a(buf, cg, 0, af[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (NAB == 2 && i + 1 < 4) load_a(buf, cg, i + 1, af[(i + 1) % NAB]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < NCOMBO; ++q) {
            const SubCombo c = sub_combo<MODE>(q);
            acc[i][c.py * 2 + c.px] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i % NAB][(c.sy + 1) * 3 + (c.sx + 1)], bf[q], acc[i][c.py * 2 + c.px], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (NAB == 1 && i + 1 < 4) load_a(buf, cg, i + 1, af[0]);
        }
      }
    }
  };

  // ---- epilogue ----
  float st_n = 0.f, st_mean[1] = {0.f}, st_m2[1] = {0.f};
  float bias_v = 0.f, mk_mu = 0.f, mk_sc = 0.f, mk_sh = 0.f;
  int bias_co0 = -1;
  auto load_bias = [&](int co0) {
    if (co0 == bias_co0) return;
    bias_co0 = co0;
    const int co = co0 + m;
    bias_v = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
    if (a.epi_mode == 3) {
      const float4 cf = (co < a.Cout) ? reinterpret_cast<const float4*>(a.mk_coef)[co] : make_float4(0.f, 0.f, 0.f, 0.f);
      mk_sc = cf.x; mk_sh = cf.y; mk_mu = cf.z;
    }
  };
  // stored coordinates of this lane's pixel quad of M-tile i: image, row, first column; false: the quad is outside
  auto quad = [&](int n_item, int tile, int i, int& n, int& y, int& x) -> bool {
    if (GEO == 0) {
      const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
      n = n_item; y = ty * TLH + 2 * wave + (i >> 1); x = tx * TLW + 4 * k + (i & 1) * 16;
      return y < a.Hs && x < a.Ws;
    }
    const int b = tile * 16 + 4 * wave + i;
    const int bx = b % nbx, t = b / nbx, by = t % nby;
    n = t / nby; y = by * 4 + k; x = bx * 4;
    return b < NB && y < a.Hs;
  };
  auto epilogue = [&](int n_item, int tile, int co0) {
    const int co = co0 + m;
    int qn[4], qy[4], qx[4], qnr[4]; bool qok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qok[i] = quad(n_item, tile, i, qn[i], qy[i], qx[i]);
      qnr[i] = (GEO == 1) ? min(4, a.Ws - qx[i]) : 4;       // stored columns of the quad inside the image (GEO 1, Ws % 4 == 2: the last block of a row holds 2)
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][p][r] += bias_v;
    if (a.stats != nullptr) {
      // per-lane running (count, mean, M2) over the outputs this lane produced (<= 64 per item), Chan-merged item by item (ms_conv_kernel.h)
      float cnt = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) if (qok[i]) cnt += 4.f * (float)qnr[i];
      if (cnt > 0.f) {
        const float rc = __builtin_amdgcn_rcpf(cnt);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += (qok[i] && r < qnr[i]) ? acc[i][p][r] : 0.f;
        }
        const float mean = s * rc;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = acc[i][p][r] - mean; qq += (qok[i] && r < qnr[i]) ? d * d : 0.f; }
        }
        const float nt_ = st_n + cnt;
        const float wgt = cnt * __builtin_amdgcn_rcpf(nt_);
        const float d = mean - st_mean[0];
        st_mean[0] += d * wgt;
        st_m2[0] += qq + d * d * st_n * wgt;
        st_n = nt_;
      }
    }
    if (co < a.Cout) {
      const int Wo = a.Wout;
      const bool have_ref = (a.epi_mode == 3 && mk_ref != nullptr);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!qok[i]) continue;
        const size_t pb = ((size_t)qn[i] * a.Cout + co) * a.Hout * Wo;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const size_t off = pb + (size_t)(2 * qy[i] + py) * Wo + 2 * qx[i];
          const f32x4 e = acc[i][py * 2], o = acc[i][py * 2 + 1];
          float4 v0 = make_float4(e[0], o[0], e[1], o[1]), v1 = make_float4(e[2], o[2], e[3], o[3]);
          const bool full = (qnr[i] == 4);                // (else 2 stored columns = the first four outputs of the row: v0)
          if (a.epi_mode == 3) {
            const float4 u0 = *reinterpret_cast<const float4*>(a.mk_u + off), u1 = full ? *reinterpret_cast<const float4*>(a.mk_u + off + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 r0, r1;
            if (have_ref) { r0 = *reinterpret_cast<const float4*>(mk_ref + off); r1 = full ? *reinterpret_cast<const float4*>(mk_ref + off + 4) : make_float4(0.f, 0.f, 0.f, 0.f); }
            else {
              r0 = make_float4(mk_sc * u0.x + mk_sh, mk_sc * u0.y + mk_sh, mk_sc * u0.z + mk_sh, mk_sc * u0.w + mk_sh);
              r1 = make_float4(mk_sc * u1.x + mk_sh, mk_sc * u1.y + mk_sh, mk_sc * u1.z + mk_sh, mk_sc * u1.w + mk_sh);
            }
            v0.x *= (r0.x > 0.f) ? 1.f : a.mk_slope; v0.y *= (r0.y > 0.f) ? 1.f : a.mk_slope; v0.z *= (r0.z > 0.f) ? 1.f : a.mk_slope; v0.w *= (r0.w > 0.f) ? 1.f : a.mk_slope;
            v1.x *= (r1.x > 0.f) ? 1.f : a.mk_slope; v1.y *= (r1.y > 0.f) ? 1.f : a.mk_slope; v1.z *= (r1.z > 0.f) ? 1.f : a.mk_slope; v1.w *= (r1.w > 0.f) ? 1.f : a.mk_slope;
            if (!full) v1 = make_float4(0.f, 0.f, 0.f, 0.f);
            s1 += ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w));
            s2 += ((v0.x * (u0.x - mk_mu) + v0.y * (u0.y - mk_mu)) + (v0.z * (u0.z - mk_mu) + v0.w * (u0.w - mk_mu))) +
                  ((v1.x * (u1.x - mk_mu) + v1.y * (u1.y - mk_mu)) + (v1.z * (u1.z - mk_mu) + v1.w * (u1.w - mk_mu)));
          }
          *reinterpret_cast<float4*>(a.out + off) = v0;
          if (full) *reinterpret_cast<float4*>(a.out + off + 4) = v1;
        }
      }
      if (a.epi_mode == 3) { st_mean[0] += s1; st_m2[0] += s2; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  int item = vb, chunk = 0, n, tile, cb, ring = 0;
  decode(item, n, tile, cb);
  load_bias(cb * 16);
  lds_barrier();                                      // barrier #0
  lds_barrier();                                      // barrier #1: chunk 0 is in buffer 0
  for (int p = 0; p < T; ++p) {
    const int c0 = chunk * CK;
    const int ncg = min(CK / 4, (a.cin_pad - c0) / 4);
    compute(smem + ring * BUF, ncg);
    if (++ring == NBUF) ring = 0;
    if (chunk + 1 == nchunks) {
      epilogue(n, tile, cb * 16);
      chunk = 0; item += gridDim.x;
      if (p + 1 < T) { decode(item, n, tile, cb); load_bias(cb * 16); }
    } else {
      ++chunk;
    }
    if (p + 1 < T) lds_barrier();
  }
  if (a.stats != nullptr) conv_table_tail<1, true>(a, smem, vb, ncb, st_n, st_mean, st_m2);
  else if (a.epi_mode == 3) conv_table_tail<1, false>(a, smem, vb, ncb, 0.f, st_mean, st_m2);
}

// geo: 0 tiles, 1 blocks, -1 choose (blocks when they fill at least 1.25x better than the tiles)
inline int subpix2_geo(const ConvArgs& a, int geo, bool mode1) {
  if (a.Ws % 4 != 0) return 1;                 // (rows of 16-byte pieces only where the width is a multiple of 4: an even width takes the blocks)
  if (geo == 0 || geo == 1) return geo;
  const double fill_t = (double)a.Hs * a.Ws / ((double)cdiv(a.Hs, 8) * 8 * cdiv(a.Ws, 32) * 32);
  const double fill_b = (double)a.Hs / (cdiv(a.Hs, 4) * 4);
  const long items_b = cdiv((long)a.N * (a.Ws / 4) * cdiv(a.Hs, 4), 16L) * cdiv(a.Cout, 16);
  const long items_t = (long)a.N * cdiv(a.Hs, 8) * cdiv(a.Ws, 32) * cdiv(a.Cout, 16);
  // (round 5, the 12 / 24-pixel levels of the reference's shipped 192-pixel workload) few, full blocks against MORE THAN ONE ROUND of mostly empty tiles: an item costs
  // the same matrix time in both geometries, two co-resident items share their SIMDs - 20 x 128 @12x12: 96 block items 100 % full against 320 tile items 28 % full
  if (items_b < num_cus() && items_t > num_cus() && fill_b >= 2.0 * fill_t) return 1;
  // measured (tools/ab_subpix.py, profiles/r04_experiments.txt 7): the blocks pay their 2.25x patch traffic in single-float DMA pieces; with mode 1's 9 products per
  // M-tile (mode 0: 16) that is worth it only where the tiles are half empty, and never when the blocks leave compute units without a work item
  return (items_b >= num_cus() && fill_b >= (mode1 ? 1.8 : 1.4) * fill_t) ? 1 : 0;
}

template <int MODE, int GEO, int NBUF>
int launch_conv_subpix2_t(ConvArgs a, const float* mk_ref, hipStream_t st) {
  using G = SubGeo2<MODE, GEO>;
  const size_t lds_bytes = sizeof(float) * NBUF * (size_t)G::BUF;
  static std::once_flag attr_once;
  std::call_once(attr_once, []() { (void)hipFuncSetAttribute((const void*)conv_subpix2_kernel<MODE, GEO, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)); });
  a.ncb = cdiv(a.Cout, 16);
  long nitems;
  if (GEO == 0) { a.tiles_x = cdiv(a.Ws, G::TLW); a.tiles_y = cdiv(a.Hs, G::TLH); nitems = (long)a.N * a.tiles_x * a.tiles_y * a.ncb; }
  else { const long nb = (long)a.N * cdiv(a.Ws, 4) * cdiv(a.Hs, 4); a.tiles_x = (int)cdiv(nb, 16L); a.tiles_y = 1; nitems = (long)a.tiles_x * a.ncb; }
  const int per_cu = std::max(1, std::min(2, (int)((160 * 1024) / (lds_bytes + 256))));
  long nblocks = std::min<long>(nitems, (long)num_cus() * per_cu);
  if (nblocks > a.ncb) nblocks -= nblocks % a.ncb;
  MS_LAUNCH((conv_subpix2_kernel<MODE, GEO, NBUF>), dim3((unsigned)nblocks), dim3(512), lds_bytes, st, a, mk_ref);
  return check_launch("conv_subpix2");
}
template <int MODE>
int launch_conv_subpix2(const ConvArgs& a, const float* mk_ref, int geo, hipStream_t st) {
  return subpix2_geo(a, geo, MODE == 1) == 1 ? launch_conv_subpix2_t<MODE, 1, 3>(a, mk_ref, st) : launch_conv_subpix2_t<MODE, 0, 3>(a, mk_ref, st);
}

}  // namespace ms
