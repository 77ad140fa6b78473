// Second generation of the 3x3 stride-1 convolution on NARROW images (rows of 12 / 14 / 16 pixels: the deepest encoder level, the code decoupler and their data-gradients -
// encoder_decoder.py:22-74, 441-445, 650-653 - at 192 / 224 / 256-pixel inputs) for gfx950.  The first generation (ms_conv_kernel.h, NARROW) runs them on 4 x 16-pixel
// tiles: at 16 pixels its staging waves spend ~280 vector + ~180 scalar instructions per 16-channel chunk on window addressing (23-28 us per
// launch for 7.7 us of matrix time), and rows that are not a multiple of 4 pixels (14: the reference's shipped Prostate workload, config/Prostate/MICCAI2022_MaxStyle.json)
// fall to its scalar staging path (62-68 us per launch, profiles/r05_step_budget_prostate224.txt).  Here:
//   * an M-tile is 16 consecutive pixels of the FLATTENED image plane (a 3x3 conv only needs each lane's own (row, column): one LDS base offset per lane and tile, every
//     per-tap / per-channel-group offset an immediate), so 12 x 12 = 9 and 16 x 16 = 16 tiles fill their MFMA rows completely and 14 x 14 = 12.25 tiles to 94 %, and a lane's
//     four accumulator rows are four consecutive pixels of one output plane: 16-byte stores whatever the row width (H*W % 4 == 0);
//   * a work item = 64 MT consecutive pixels of one image (MT = 1 | 2 M-tiles per MFMA wave) x 16 output channels; its input is the BAND of image rows those pixels touch
//     plus the halo, staged per 16-channel chunk into [16][PS] (row stride W + 2, plane stride == 16 mod 32 banks: conflict-free A fragments);
//   * PRO 0: the band and the weight slice travel to LDS by LDS-DMA - dword pieces for the band (64 consecutive LDS dwords from 64 per-lane global offsets: any row width,
//     zeros for halo / out-of-image positions from the buffer range check), 16-byte pieces for the weights - nothing through registers, three stage buffers;
//     PRO 1 / 2 (BatchNorm apply + LeakyReLU / BatchNorm backward of the producer layer): a staging thread owns ONE channel of the chunk (its coefficients: one LDS read per
//     chunk) and 12 fixed band positions: buffer loads with hoisted per-item offsets -> 2-4 vector instructions per element -> dword LDS stores at immediate offsets;
//   * 16-channel chunks, taps outer, 4-channel groups inner: the first generation's accumulation order on these layers (NT = 1, CK = 16) - same bits in `out`; the
//     statistics / activation-backward tables agree to summation order (per-lane Chan merge of another pixel grouping).
// Epilogues: bias, plain store | accumulate (epi_mode 1) | BatchNorm statistics | activation backward + BatchNorm-backward sums (epi_mode 3, ms_conv2d_actbwd); prologue
// coefficients from a table or derived in the launch (`_xfin`, kinds 0 and 1).  fp32 storage, per-channel coefficients, Cin % 16 == 0.
#pragma once
#include <cstdlib>
#include <map>
#include <mutex>
#include "ms_conv_kernel.h"

namespace ms {

template <int W, int MT, int KS = 3, int S = 1>
struct K3nGeo {
  static constexpr int CK = (S == 1) ? 16 : 8;                          // (stride 2: the band is four times the pixels - 8-channel chunks keep three stage buffers in LDS)
  static constexpr int TAPS = KS * KS;                                  // 9 | 1 (round 5: the 1x1 convs on 14-pixel rows share the kernel - same band, centre tap only)
  static constexpr int WI = S * W;                                      // input row width (S = 2: the stride-2 conv onto rows of W pixels, round 5)
  static constexpr int RS = WI + 2;
  static constexpr int PIX = 64 * MT;                                   // pixels of a work item: 4 MFMA waves x MT M-tiles x 16
  static constexpr int SPAN = (PIX % W == 0) ? PIX / W : (W - 1 + PIX - 1) / W + 1;      // image rows PIX consecutive pixels can touch (item starts are multiples of PIX)
  static constexpr int BR = S * (SPAN - 1) + 3;                         // input rows of the band: S (SPAN - 1) + 1 plus the halo rows
  static constexpr int BASE = BR * RS;
  static constexpr int PS = BASE + ((16 - BASE % 32 + 32) % 32);        // plane stride == 16 (mod 32 banks)
  static constexpr int IN_FLOATS = CK * PS;
  static constexpr int NJ = (IN_FLOATS + 255) / 256;                    // dword DMA pieces per staging wave and chunk
  static constexpr int IN_REGION = NJ * 256;
  static constexpr int W_FLOATS = TAPS * CK * 16;                       // [tap][channel][16 output channels]
  static constexpr int NWJ = (W_FLOATS / 4 + 255) / 256;                // 16-byte DMA pieces per staging wave and chunk (3; the last round half used)
  static constexpr int W_REGION = NWJ * 1024;
  static constexpr int BUF = IN_REGION + W_REGION;                      // floats per stage buffer
  static constexpr int NE = (BASE + 15) / 16;                           // register path: band positions per staging thread (16 threads per channel)
  static constexpr int OOB = (int)0x80000000;
};

// PRO: 0 none | 1 v = lrelu(a[c] v + b[c]) | 2 v = a[c] v + b[c] v2 + c[c]
template <int W, int MT, int PRO, bool CHAINED = false, int KS = 3, int S = 1>
__device__ __forceinline__ void conv_k3n_body(const ConvArgs& a, float* smem) {
  static_assert(S == 1 || (PRO == 0 && KS == 3), "the stride-2 form is the prologue-free 3x3 conv");
  using G = K3nGeo<W, MT, KS, S>;
  constexpr int WI = G::WI;
  constexpr int CK = G::CK, RS = G::RS, PS = G::PS, PIX = G::PIX, BUF = G::BUF, NJ = G::NJ, NWJ = G::NWJ, NE = G::NE, OOB = G::OOB;
  float* cf_lds = smem + 3 * BUF;                        // [cin_pad][4] prologue coefficients
  const int wave = MS_TID >> 6, lane = MS_TID & 63;
  const bool producer = wave >= 4;
  const int ncb = a.ncb;
  const int HW = a.Hout * a.Wout, HWI = a.Hs * a.Ws;     // output / input plane (equal at stride 1)
  const int gpi = (HW + PIX - 1) / PIX;                  // pixel groups per image
  const int nitems = a.N * gpi * ncb;
  const int nchunks = a.cin_pad / CK;
  const int vb = ((int)gridDim.x % 8 == 0) ? (((int)blockIdx.x % 8) * ((int)gridDim.x / 8) + (int)blockIdx.x / 8) : (int)blockIdx.x;
  const int my_items = (nitems - vb + (int)gridDim.x - 1) / (int)gridDim.x;
  const int T = my_items * nchunks;
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto decode = [&](int it, int& n, int& grp, int& cb) { cb = it % ncb; const int t2 = it / ncb; grp = t2 % gpi; n = t2 / gpi; };

  if (PRO != 0 && a.xf_tab == nullptr) {
    for (int c = MS_TID; c < a.cin_pad; c += 512) {
      float ca = 1.f, cb_ = 0.f, cc = 0.f;
      if (c < a.Cin) { ca = a.pro_a[c * a.pro_cstride]; cb_ = a.pro_b[c * a.pro_cstride]; if (PRO == 2) cc = a.pro_c[c * a.pro_cstride]; }
      reinterpret_cast<float4*>(cf_lds)[c] = make_float4(ca, cb_, cc, 0.f);
    }
  }

  if (producer) {
    // =========================================== STAGING waves ===========================================
    __builtin_amdgcn_s_setprio(3);
    const int sw = __builtin_amdgcn_readfirstlane(wave) - 4;
    const ms_i32x4 rs_w = ms_dma_rsrc_n(a.w, (unsigned)G::TAPS * a.cin_pad * a.cout_pad * 4u);
    const unsigned lds0 = ms_lds_addr(smem);
    // weights: 16-byte piece q = (sw + 4 j) * 64 + lane -> row (tap, c) = q / 4, part = q % 4
    int w_voff[NWJ];
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
      const int q = (sw + 4 * j) * 64 + lane;
      const int row = q >> 2, part = q & 3, tap = row / CK, c = row - tap * CK;
      w_voff[j] = (q < G::W_FLOATS / 4) ? (int)((((size_t)tap * a.cin_pad + c) * a.cout_pad + 4 * part) * 4) : OOB;
    }
    auto issue_w = [&](int buf, int cb, int chunk) {
      const unsigned lb = lds0 + (unsigned)buf * (BUF * 4) + G::IN_REGION * 4;
#pragma unroll
      for (int j = 0; j < NWJ; ++j) ms_lds_dma16(rs_w, lb + (unsigned)(sw + 4 * j) * 1024, w_voff[j], ((chunk * CK) * a.cout_pad + cb * 16) * 4);
    };
    int item = vb, chunk = 0, n, grp, cb, ring = 0;
    decode(item, n, grp, cb);

    if constexpr (PRO == 0) {
      // ---- LDS-DMA: dword piece (sw + 4 j): LDS dwords L = ((sw + 4 j) * 64 + lane) of the [CK][PS] band image
      const ms_i32x4 rs_in = ms_dma_rsrc_n(a.in, (unsigned)a.N * a.Cin * HWI * 4u);
      int i_sb[NJ], i_br[NJ], i_voff[NJ];        // static byte offset (channel, band row, column) | band row (or -1000: no element) | per-item offset
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int L = (sw + 4 * j) * 64 + lane;
        const int c = L / PS, rem = L - c * PS, br = rem / RS, bc = rem - br * RS;
        const bool ok = (c < CK) && (rem < G::BASE) && (bc >= 1) && (bc <= WI);
        i_sb[j] = (c * HWI + br * WI + (bc - 1)) * 4;
        i_br[j] = ok ? br : -1000;
        i_voff[j] = OOB;
      }
      auto set_item = [&](int grp_) {
        const int r0 = (grp_ * PIX) / W;                    // output row of the item's first pixel; band row 0 = input row S r0 - 1
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int row = S * r0 - 1 + i_br[j];
          i_voff[j] = ((unsigned)row < (unsigned)a.Hs) ? (i_sb[j] + (S * r0 - 1) * WI * 4) : OOB;
        }
      };
      auto issue = [&](int buf, int n_, int cb_, int chunk_) {
        issue_w(buf, cb_, chunk_);
        const unsigned lb = lds0 + (unsigned)buf * (BUF * 4);
        const int soff = (n_ * a.Cin + chunk_ * CK) * HWI * 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) ms_lds_dma4(rs_in, lb + (unsigned)(sw + 4 * j) * 256, i_voff[j], soff);
      };
      int grp_set = -1;
      auto issue_next = [&](bool more) {
        if (grp != grp_set) { set_item(grp); grp_set = grp; }
        issue(ring, n, cb, chunk);
        if (++ring == 3) ring = 0;
        if (++chunk == nchunks) { chunk = 0; item += gridDim.x; if (more) decode(item, n, grp, cb); }
      };
      lds_barrier();                                    // barrier #0
      issue_next(T > 1);
      for (int p = 0; p < T; ++p) {
        if (p + 1 < T) { issue_next(p + 2 < T); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ + NWJ) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                                  // barrier #(p+1): chunk p has landed
      }
      return;
    } else {
      // ---- register path: thread t of the 256 owns channel c = t / 16 of the chunk and band positions e = (t % 16) + 16 i
      const int tid = MS_TID - 256;
      const int c = tid >> 4, e0 = tid & 15;
      const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((unsigned)a.N * a.Cin * HW * 4u), 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRO == 2 ? a.in2 : a.in), 0, (int)((unsigned)a.N * a.Cin * HW * 4u), 0x00020000);
      int e_sb[NE], e_br[NE], e_voff[NE];
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int e = e0 + 16 * i;
        const int br = e / RS, bc = e - br * RS;
        const bool ok = (e < G::BASE) && (bc >= 1) && (bc <= W);
        e_sb[i] = (c * HW + br * W + (bc - 1)) * 4;
        e_br[i] = ok ? br : -1000;
        e_voff[i] = OOB;
      }
      unsigned okmask = 0u;
      auto set_item = [&](int grp_) {
        const int r0 = (grp_ * PIX) / W;
        okmask = 0u;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
          const int row = r0 - 1 + e_br[i];
          const bool ok = (unsigned)row < (unsigned)a.Hs;
          e_voff[i] = ok ? (e_sb[i] + (r0 - 1) * W * 4) : OOB;
          okmask |= ok ? (1u << i) : 0u;
        }
      };
      float rv[NE], rv2[PRO == 2 ? NE : 1];
      unsigned rmask = 0u;                               // okmask of the item whose values sit in rv
      int r_c0 = 0;                                      // first channel of the chunk in rv
      auto load = [&](int n_, int chunk_) {
        const int soff = (n_ * a.Cin + chunk_ * CK) * HW * 4;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
          rv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, e_voff[i], soff, 0));
          if constexpr (PRO == 2) rv2[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in2, e_voff[i], soff, 0));
        }
        rmask = okmask; r_c0 = chunk_ * CK;
      };
      auto store = [&](int buf) {
        float* dst = smem + (size_t)buf * BUF + c * PS + e0;
        const float4 cf = reinterpret_cast<const float4*>(cf_lds)[r_c0 + c];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
          if (16 * i + 15 >= G::BASE) { if (e0 + 16 * i >= G::BASE) continue; }      // (only the last round can leave the band)
          float v = rv[i];
          if constexpr (PRO == 1) v = leaky(cf.x * v + cf.y, a.slope);
          if constexpr (PRO == 2) v = cf.x * v + cf.y * rv2[i] + cf.z;      // (the first generation's association: (a v + b v2) + c)
          v = ((rmask >> i) & 1u) ? v : 0.f;             // zero padding pads the tensor AFTER the prologue
          dst[16 * i] = v;
        }
      };
      int grp_set = -1;
      int l_n = n, l_grp = grp, l_cb = cb, l_chunk = 0, l_item = item;      // cursor of the LOAD stream
      int w_cb = cb, w_chunk = 0, w_item = item;                            // cursor of the WRITE stream (one chunk behind the loads)
      auto adv = [&](int& it_, int& ch_, int& n_, int& g_, int& cb_, bool more) {
        if (++ch_ == nchunks) { ch_ = 0; it_ += gridDim.x; if (more) decode(it_, n_, g_, cb_); }
      };
      // chunk 0: loads, then (behind barrier #0: the coefficient table is complete) its LDS image; chunk 1's loads in flight behind it
      if (l_grp != grp_set) { set_item(l_grp); grp_set = l_grp; }
      load(l_n, l_chunk);
      lds_barrier();                                    // barrier #0
      issue_w(0, w_cb, w_chunk);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWJ) : "memory");
      store(0);
      ring = 1;
      {
        int dn, dg;
        adv(w_item, w_chunk, dn, dg, w_cb, T > 1);
      }
      adv(l_item, l_chunk, l_n, l_grp, l_cb, T > 1);
      if (T > 1) { if (l_grp != grp_set) { set_item(l_grp); grp_set = l_grp; } load(l_n, l_chunk); }
      for (int p = 0; p < T; ++p) {
        if (p + 1 < T) {
          // chunk p + 1: weights by DMA, the band from the registers loaded one iteration ago; then chunk p + 2's loads
          issue_w(ring, w_cb, w_chunk);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWJ) : "memory");
          store(ring);
          if (++ring == 3) ring = 0;
          {
            int dn, dg;
            adv(w_item, w_chunk, dn, dg, w_cb, p + 2 < T);
          }
          adv(l_item, l_chunk, l_n, l_grp, l_cb, p + 2 < T);
          if (p + 2 < T) { if (l_grp != grp_set) { set_item(l_grp); grp_set = l_grp; } load(l_n, l_chunk); }
          // the weight DMA of chunk p + 1 must have landed before the barrier that publishes it; the loads of chunk p + 2 stay in flight
          if (p + 2 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PRO == 2 ? 2 : 1) * NE) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_barrier();                                  // barrier #(p+1)
      }
      return;
    }
  }

  // =========================================== MFMA waves ===========================================
  unsigned xf_tag = 0u;
  int xf_nparts = 0;
  if (a.xf_tab != nullptr) {
    // the coefficients feed this launch's PROLOGUE: in LDS before the staging waves write their first chunk (barrier #0)
    xfin_header(a, xf_tag, xf_nparts);
    if (!xfin_produce(a, xf_tag, xf_nparts)) __builtin_amdgcn_s_sleep(30);
    if (PRO == 2) xfin_fill<3>(a, cf_lds, a.cin_pad, xf_tag, vb & (kXfinRep - 1), MS_TID, 256, 1.f, 0.f);
    else xfin_fill<2>(a, cf_lds, a.cin_pad, xf_tag, vb & (kXfinRep - 1), MS_TID, 256, 1.f, 0.f);
  }
  const int m = lane & 15, k = lane >> 4;
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_off[MT];
  const int b_lane = G::IN_REGION + k * 16 + m;
  auto set_item = [&](int grp) {
    const int p0 = grp * PIX, r0 = p0 / W;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int pix = min(p0 + (wave * MT + t) * 16 + m, HW - 1);
      const int y = pix / W, x = pix - y * W;
      a_off[t] = k * PS + S * (y - r0) * RS + S * x;
    }
  };
  auto compute = [&](const float* buf) {
#pragma unroll
    for (int tap = 0; tap < G::TAPS; ++tap) {
#pragma unroll
      for (int cg = 0; cg < CK / 4; ++cg) {
        const float bf = buf[b_lane + (tap * CK + cg * 4) * 16];
        float af[MT];
        constexpr int HO = (3 - KS) / 2;                 // (KS 1: the centre of the 3 x 3 window the band is laid out for)
#pragma unroll
        for (int t = 0; t < MT; ++t) af[t] = buf[a_off[t] + cg * 4 * PS + (tap / KS + HO) * RS + (tap % KS + HO)];
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t], bf, acc[t], 0, 0, 0);
      }
    }
  };

  float st_n = 0.f, st_mean[1] = {0.f}, st_m2[1] = {0.f};
  float bias_v = 0.f, mk_sc = 0.f, mk_sh = 0.f, mk_mu = 0.f;
  int cur_cb = -1;
  auto load_cb = [&](int cb) {
    if (cb == cur_cb) return;
    cur_cb = cb;
    const int co = cb * 16 + m;
    bias_v = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
    if (a.epi_mode == 3) {
      const float4 cf = (co < a.Cout) ? reinterpret_cast<const float4*>(a.mk_coef)[co] : make_float4(0.f, 0.f, 0.f, 0.f);
      mk_sc = cf.x; mk_sh = cf.y; mk_mu = cf.z;
    }
  };
  // D layout: this lane holds output channel m of pixels (tile base) + 4 k .. + 3: four consecutive floats of one output plane
  auto epilogue = [&](int n, int grp, int cb) {
    const int co = cb * 16 + m;
    const int p0 = grp * PIX;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] += bias_v;
    if (a.stats != nullptr) {
      float cnt = 0.f;
#pragma unroll
      for (int t = 0; t < MT; ++t) cnt += (float)max(0, min(4, HW - (p0 + (wave * MT + t) * 16 + 4 * k)));
      if (cnt > 0.f) {
        const float rc = __builtin_amdgcn_rcpf(cnt);
        const float nt_ = st_n + cnt;
        const float wgt = cnt * __builtin_amdgcn_rcpf(nt_);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) s += (p0 + (wave * MT + t) * 16 + 4 * k + r < HW) ? acc[t][r] : 0.f;
        const float mean = s * rc;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = acc[t][r] - mean; q += (p0 + (wave * MT + t) * 16 + 4 * k + r < HW) ? d * d : 0.f; }
        const float d = mean - st_mean[0];
        st_mean[0] += d * wgt;
        st_m2[0] += q + d * d * st_n * wgt;
        st_n = nt_;
      }
    }
    const size_t pb = ((size_t)n * a.Cout + min(co, a.Cout - 1)) * (size_t)HW;
    if (a.epi_mode == 3) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int px = p0 + (wave * MT + t) * 16 + 4 * k;
        if (co < a.Cout && px < HW) {                     // (H*W % 4 == 0: a quad is inside the plane or outside it)
          const float4 uu = *reinterpret_cast<const float4*>(a.mk_u + pb + px);
          float4 v;
          v.x = acc[t][0] * ((mk_sc * uu.x + mk_sh > 0.f) ? 1.f : a.mk_slope); v.y = acc[t][1] * ((mk_sc * uu.y + mk_sh > 0.f) ? 1.f : a.mk_slope);
          v.z = acc[t][2] * ((mk_sc * uu.z + mk_sh > 0.f) ? 1.f : a.mk_slope); v.w = acc[t][3] * ((mk_sc * uu.w + mk_sh > 0.f) ? 1.f : a.mk_slope);
          *reinterpret_cast<float4*>(a.out + pb + px) = v;
          s1 += (v.x + v.y) + (v.z + v.w);
          s2 += (v.x * (uu.x - mk_mu) + v.y * (uu.y - mk_mu)) + (v.z * (uu.z - mk_mu) + v.w * (uu.w - mk_mu));
        }
      }
      st_mean[0] += s1; st_m2[0] += s2;
    } else {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int px = p0 + (wave * MT + t) * 16 + 4 * k;
        if (co < a.Cout && px < HW) {
          float4 v = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
          if (a.epi_mode == 1) { const float4 p = *reinterpret_cast<const float4*>(a.out + pb + px); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
          *reinterpret_cast<float4*>(a.out + pb + px) = v;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  int item = vb, chunk = 0, n, grp, cb, ring = 0;
  decode(item, n, grp, cb);
  load_cb(cb);
  set_item(grp);
  lds_barrier();                                      // barrier #0
  lds_barrier();                                      // barrier #1: chunk 0 is in buffer 0
  for (int p = 0; p < T; ++p) {
    if (!(a.dbg & 1)) compute(smem + ring * BUF);
    if (++ring == 3) ring = 0;
    if (chunk + 1 == nchunks) {
      if (!(a.dbg & 4)) epilogue(n, grp, cb);
      chunk = 0; item += gridDim.x;
      if (p + 1 < T) { decode(item, n, grp, cb); load_cb(cb); set_item(grp); }
    } else {
      ++chunk;
    }
    if (p + 1 < T) lds_barrier();
  }
  if constexpr (!CHAINED) {      // (the table tail synchronises the MFMA waves only: the staging waves of a stand-alone launch have returned)
    if (a.stats != nullptr) conv_table_tail<1, true>(a, smem, vb, ncb, st_n, st_mean, st_m2);
    else if (a.epi_mode == 3) conv_table_tail<1, false>(a, smem, vb, ncb, 0.f, st_mean, st_m2);
  }
}

template <int W, int MT, int PRO, int KS = 3, int S = 1>
__global__ __launch_bounds__(512, 4) void conv_k3n_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv_k3n_body<W, MT, PRO, false, KS, S>(a, smem);
}

// ---- go / no-go probe for a LAYER-CHAIN launch: L plain 3x3 layers (prologue-free, plain store) walked by ONE persistent
// launch - the body above per layer, then a grid barrier (every wave drains its stores, one agent-scope release + arrival per workgroup, a bounded spin on the arrival
// counter, an agent-scope acquire) instead of a launch boundary.  Diagnostics only (ms_diag_k3n_chain, tools/chain_probe.py): no product path launches it.
template <int W, int MT>
__global__ __launch_bounds__(512, 4) void conv_k3n_chain_kernel(const ConvArgs* __restrict__ layers, int L, unsigned* __restrict__ arrive, int* __restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned base = __hip_atomic_load(arrive + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // arrivals of earlier launches (the host never resets the counter)
  for (int l = 0; l < L; ++l) {
    const ConvArgs a = layers[l];
    conv_k3n_body<W, MT, 0, true>(a, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MS_TID == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);            // (agent scope: the workgroup's stores leave this XCD's L2)
      __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = base + (unsigned)(l + 1) * gridDim.x;
      for (unsigned spins = 0;; ++spins) {
        if ((int)(__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) break;
        if (spins > (1u << 20)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(2);
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && MS_TID == 0) {
    // the last workgroup through would be cleaner; block 0 has seen every arrival of the last barrier: publish the new base for the next launch
    __hip_atomic_store(arrive + 1, base + (unsigned)L * gridDim.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Eligible: 3x3 stride 1, plain fetch, fp32 storage, rows of 12 / 14 / 16 pixels, whole 16-channel chunks, per-channel coefficients, 16-byte pixel quads
// stride 2 (round 5: res_convdown.down onto rows of 12 / 14 / 16 pixels - encoder_decoder.py:40 at the deepest level; the DMA-staged stride-2 kernel needs 16-byte output
// quads and 32-pixel tiles, the first generation ran it on its scalar path at 14 pixels: 44.9 us at the shipped Prostate shape): prologue-free, bias, plain store
inline bool conv_k3n_s2_eligible(const ConvArgs& a, int ks, int stride, int fetch) {
  if (opt(OPT_CONV_K3N) == 0 || ks != 3 || stride != 2 || fetch != FETCH_NORMAL || a.act_bf16 != 0 || a.pro_mode != 0 || a.in2 != nullptr || a.stats != nullptr ||
      a.epi_mode != 0 || a.xf_tab != nullptr || a.ride_out != nullptr) return false;
  if (!(a.Wout == 12 || a.Wout == 14 || a.Wout == 16) || a.Ws != 2 * a.Wout || a.Hs != 2 * a.Hout || (a.Hout * a.Wout) % 4 != 0) return false;
  if (a.Cin % 8 != 0 || a.cin_pad != a.Cin || a.Cin < 16) return false;
  if ((long long)a.N * a.Cin * a.Hs * a.Ws * 4 >= (1LL << 31) || 9LL * a.cin_pad * a.cout_pad * 4 >= (1LL << 31)) return false;
  return aligned16(a.in) && aligned16(a.out) && aligned16(a.w);
}
inline bool conv_k3n_eligible(const ConvArgs& a, int ks, int stride, int fetch) {
  if (stride == 2) return conv_k3n_s2_eligible(a, ks, stride, fetch);
  if (opt(OPT_CONV_K3N) == 0 || !(ks == 3 || ks == 1) || stride != 1 || fetch != FETCH_NORMAL || a.act_bf16 != 0) return false;
  // 1x1: only rows of 14 pixels (not a multiple of 4: the tiled first-generation 1x1 falls to its scalar staging there - 28-33 us per launch at the shipped Prostate shape
  // against ~8 us at 12 / 16 pixels) and only the plain / statistics / two-tensor-prologue calls (the residual tails and riders have the GEMM kernel, ms_conv_k1g.h)
  if (ks == 1 && (a.Ws != 14 || a.epi_mode == 3 || a.Cin < 64)) return false;
  if (!(a.Ws == 12 || a.Ws == 14 || a.Ws == 16) || a.Hs < 1 || (a.Hs * a.Ws) % 4 != 0) return false;
  if (a.Cin % 16 != 0 || a.cin_pad != a.Cin || a.Cin < 16) return false;
  if (a.pro_mode != 0 && a.pro_nstride != 0) return false;
  if (!(a.epi_mode == 0 || a.epi_mode == 1 || a.epi_mode == 3) || a.ride_out != nullptr) return false;
  if (a.xf_tab != nullptr && a.pro_mode == 0) return false;
  if ((long long)a.N * a.Cin * a.Hs * a.Ws * 4 >= (1LL << 31) || 9LL * a.cin_pad * a.cout_pad * 4 >= (1LL << 31)) return false;
  if ((size_t)a.cin_pad * 16 + 3 * sizeof(float) * (size_t)K3nGeo<16, 2>::BUF > 150 * 1024) return false;      // coefficient table beside three stage buffers
  return aligned16(a.in) && aligned16(a.out) && aligned16(a.w) && (a.in2 == nullptr || aligned16(a.in2)) && (a.epi_mode != 3 || aligned16(a.mk_u));
}

// M-tiles per MFMA wave.  Two halve the weight traffic and stage 25 % instead of 50 % halo rows per band - taken where they leave a work item for every CU AND do not
// cost matrix time: an image of HW pixels is ceil(HW / (64 MT)) items of MT units each (12 x 12 = 144 pixels: 3 x 1 units against 2 x 2; 14 x 14 and 16 x 16: equal).
// (Measured and NOT adopted, round 5: one tile at 20 x 128 @14x14, where two tiles mean 320 items = two rounds on 256 CUs against 640 half-size items = three - the
//  launches went from 30-32 to 32-35 us: the doubled weight and halo traffic costs more than the idle quarter of the second round.)
inline int conv_k3n_mt(const ConvArgs& a) {
  const int HW = a.Hout * a.Wout;
  const long items2 = (long)a.N * cdiv(HW, 128) * cdiv(a.Cout, 16);
  if (items2 < (long)num_cus()) return 1;
  return (2 * cdiv(HW, 128) <= cdiv(HW, 64)) ? 2 : 1;
}

template <int W, int MT, int PRO, int KS = 3, int S = 1>
int launch_conv_k3n_t(ConvArgs a, hipStream_t st) {
  using G = K3nGeo<W, MT, KS, S>;
  const size_t lds_bytes = sizeof(float) * (3 * (size_t)G::BUF + 4 * (size_t)a.cin_pad);
  static std::once_flag attr_once;
  std::call_once(attr_once, []() { (void)hipFuncSetAttribute((const void*)conv_k3n_kernel<W, MT, PRO, KS, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)); });
  a.ncb = cdiv(a.Cout, 16);
  const long nitems = (long)a.N * cdiv(a.Hout * a.Wout, G::PIX) * a.ncb;
  const int per_cu = std::max(1, std::min(conv_resident_per_cu((const void*)conv_k3n_kernel<W, MT, PRO, KS, S>, lds_bytes), 2));
  long nblocks = std::min<long>(nitems, (long)num_cus() * per_cu);
  if (nblocks > a.ncb) nblocks -= nblocks % a.ncb;
  MS_LAUNCH((conv_k3n_kernel<W, MT, PRO, KS, S>), dim3((unsigned)nblocks), dim3(512), lds_bytes, st, a);
  return check_launch("conv_k3n");
}

int conv_dispatch_k3n(const ConvArgs& a, int ks, hipStream_t st, int stride = 1);      // ms_conv_inst_k3n.hip

}  // namespace ms
