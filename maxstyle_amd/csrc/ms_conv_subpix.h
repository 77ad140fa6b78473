// Sub-pixel ("parity-decomposed") convolution kernel for gfx950: the two x2 resampling convolutions of the networks WITHOUT multiplying by the
// structural zeros / duplicates the resampling inserts.
//
//   MODE 0  nn.UpsamplingNearest2d(2) -> nn.Conv2d(3x3, p=1)          (up_type 'NN' blocks, encoder_decoder.py:298-300, 323-337)
//           out[2y+py, 2x+px] = sum_{ty,tx in {0,1}} W'[py,px,ty,tx] . in[y+py-1+ty, x+px-1+tx]
//           W' = the 3x3 taps that fall on the same stored pixel, summed: rows {0 | 1+2} for py = 0, {0+1 | 2} for py = 1 (same for columns).
//           4 taps per output pixel instead of 9: 2.25x fewer MFMAs than convolving the up-sampled tensor (the first-generation kernel's FETCH_UPS2).
//   MODE 1  data-gradient of nn.Conv2d(3x3, s=2, p=1)                  (res_convdown.down, encoder_decoder.py:40)
//           din[2y+py, 2x+px] = sum_{valid (ky,kx)} W[ky,kx]^T . g[y+sy, x+sx],  ky = 1 (py = 0) | 0 with sy = 1, 2 with sy = 0 (py = 1)
//           9 (parity, tap) products per 2x2 output block instead of 36: 4x fewer MFMAs than the zero-insertion form (FETCH_ZINS2).
// The fp32 MFMA runs at the fp32 VECTOR rate and does not co-issue with vector instructions (tools/probes/coexec_probe.hip: matrix + vector time
// add up), so multiplying zeros is never free on this chip.
//
// Both modes read the STORED (low-resolution) tensor and write the high-resolution one; the weights are the ordinary packed layouts
// (forward [tap][cin_pad][cout_pad] for MODE 0, data-gradient layout for MODE 1) - the tap sums of MODE 0 are formed while staging the weight slice
// into LDS, so there is no second packed copy to keep in step with the optimiser.
//
// Structure = the first-generation kernel's (ms_conv_kernel.h): persistent 512-thread workgroups, waves 4-7 stage (16-byte loads -> LDS, double
// buffered, one LDS-only barrier per 8-channel chunk), waves 0-3 multiply: a low-resolution tile of 8 x 32 pixels (16 x 64 outputs) per item, wave w
// owns stored rows 2w, 2w+1 = four 16-pixel M-tiles, four parity accumulators each; an A fragment (16 pixels x 4 channels, one ds_read_b32) is
// shared by every parity that uses its shift.  Epilogue: +bias, [BatchNorm statistics of the outputs], the two column parities of a lane are
// interleaved in registers -> 16-byte stores; or (MODE 1) the activation-backward mask + BatchNorm-backward sums of the layer below
// (what ms_act_bwd_reduce does in its own pass over the result).
#pragma once
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include "ms_conv_kernel.h"

namespace ms {

template <int MODE>
struct SubGeo {
  static constexpr int TLH = 8, TLW = 32, CK = 8;
  static constexpr int IH = TLH + 2;                         // stored rows y0-1 .. y0+8
  static constexpr int RS = 40;                              // stored columns x0-4 .. x0+35 (16-byte aligned window)
  static constexpr int PS = IH * RS;                         // 400 == 16 (mod 32): the four channel planes of an A fragment hit disjoint banks
  static constexpr int NCOMBO = (MODE == 0) ? 16 : 9;        // (parity, tap) products per 4-channel group
  static constexpr int WS = 16;
  static constexpr int BUF = CK * PS + NCOMBO * CK * WS;
  static constexpr int ITEMS = CK * IH * (RS / 4);
  static constexpr int NI = (ITEMS + 255) / 256;
  static constexpr int W_ITEMS = NCOMBO * CK * 4;
  static constexpr int NWI = (W_ITEMS + 255) / 256;
};

// (parity, tap) enumeration.  q -> output parity (py, px), shift (sy, sx) in {-1,0,1} of the stored pixel, and the packed-weight taps that feed it.
struct SubCombo { int py, px, sy, sx, ky0, ky1, kx0, kx1; };     // ky0..ky1 / kx0..kx1: inclusive ranges of 3x3 taps summed into this product
template <int MODE>
__host__ __device__ constexpr SubCombo sub_combo(int q) {
  if (MODE == 0) {
    const int tx = q & 1, ty = (q >> 1) & 1, px = (q >> 2) & 1, py = (q >> 3) & 1;
    // rows: py = 0: ty 0 <- ky {0}, ty 1 <- ky {1,2};  py = 1: ty 0 <- ky {0,1}, ty 1 <- ky {2}
    const int ky0 = (py == 0) ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = (py == 0) ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
    const int kx0 = (px == 0) ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = (px == 0) ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
    return SubCombo{py, px, py - 1 + ty, px - 1 + tx, ky0, ky1, kx0, kx1};
  }
  // MODE 1: q = 0: (0,0); 1,2: (0,1) sx = 0,1; 3,4: (1,0) sy = 0,1; 5..8: (1,1) (sy,sx) = (0,0),(0,1),(1,0),(1,1)
  int py = 0, px = 0, sy = 0, sx = 0;
  if (q == 0) { }
  else if (q <= 2) { px = 1; sx = q - 1; }
  else if (q <= 4) { py = 1; sy = q - 3; }
  else { py = 1; px = 1; sy = (q - 5) >> 1; sx = (q - 5) & 1; }
  // forward tap of the stride-2 conv: ky = 1 (py = 0) | 0 (sy = 1) | 2 (sy = 0); the data-gradient layout stores it flipped: tap' = 2 - k
  const int ky = (py == 0) ? 1 : (sy == 1 ? 0 : 2), kx = (px == 0) ? 1 : (sx == 1 ? 0 : 2);
  return SubCombo{py, px, sy, sx, 2 - ky, 2 - ky, 2 - kx, 2 - kx};
}
template <int MODE>
__host__ __device__ constexpr bool sub_shift_used(int sy, int sx) {
  for (int q = 0; q < SubGeo<MODE>::NCOMBO; ++q) { const SubCombo c = sub_combo<MODE>(q); if (c.sy == sy && c.sx == sx) return true; }
  return false;
}

// a.Hs, a.Ws: stored (low-resolution) size; a.Hout = 2*Hs, a.Wout = 2*Ws.  a.epi_mode: 0 plain, 3 activation-backward mask against the MATERIALISED
// activation a.mk_ref (sign(out) == sign(pre-activation)) with the sums of g and g*(mk_u - mean) for the BatchNorm backward of mk_u's layer.
template <int MODE, typename AT = float>
__global__ __launch_bounds__(512, 4) void conv_subpix_kernel(const ConvArgs a, const float* __restrict__ mk_ref) {
  using G = SubGeo<MODE>;
  using IO = ActIO<AT>;                     // storage type of the activation tensors (in, out, mk_u, mk_ref): float | ms_bf16
  constexpr int TLH = G::TLH, TLW = G::TLW, CK = G::CK, IH = G::IH, RS = G::RS, PS = G::PS, NCOMBO = G::NCOMBO, WS = G::WS, BUF = G::BUF;
  constexpr int NI = G::NI, NWI = G::NWI;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = MS_TID >> 6, lane = MS_TID & 63;
  const bool producer = wave >= 4;
  const int ntiles = a.tiles_x * a.tiles_y, ncb = a.ncb;
  const int nitems = a.N * ntiles * ncb;
  const int nchunks = (a.cin_pad + CK - 1) / CK;
  const int vb = ((int)gridDim.x % 8 == 0) ? (((int)blockIdx.x % 8) * ((int)gridDim.x / 8) + (int)blockIdx.x / 8) : (int)blockIdx.x;
  const int my_items = (nitems - vb + (int)gridDim.x - 1) / (int)gridDim.x;
  const int T = my_items * nchunks;
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto decode = [&](int it, int& n, int& tile, int& cb) { cb = it % ncb; const int t2 = it / ncb; tile = t2 % ntiles; n = t2 / ntiles; };

  if (producer) {
    // =========================================== STAGING waves ===========================================
    __builtin_amdgcn_s_setprio(3);
    const int tid = MS_TID - 256;
    const int plane = a.Hs * a.Ws;
    int s_lds[NI], s_rc[NI];                // LDS float offset | (channel << 20), (row << 16) | column-quad; -1: no item
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int it = tid + j * 256;
      s_lds[j] = -1; s_rc[j] = 0;
      if (it < G::ITEMS) {
        const int f = it % (RS / 4), row = it / (RS / 4);
        const int r = row % IH, c = row / IH;
        s_lds[j] = (c << 20) | (c * PS + r * RS + 4 * f);
        s_rc[j] = (r << 16) | f;
      }
    }
    int s_goff[NI];
    auto set_tile = [&](int tile) {
      const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
      const int y0 = ty * TLH - 1, x0 = tx * TLW - 4;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int Y = y0 + (s_rc[j] >> 16), X = x0 + 4 * (s_rc[j] & 0xFFFF);
        const bool ok = (s_lds[j] >= 0) && (Y >= 0) && (Y < a.Hs) && (X >= 0) && (X < a.Ws);      // Ws % 4 == 0: a quad is inside or outside as a whole
        s_goff[j] = ok ? (Y * a.Ws + X) : -1;
      }
    };
    float4 rin[NI], rw[NWI];
    bool have_w = false;
    auto load_chunk = [&](int n, int co0, int c0, bool load_w) {
      const size_t in_n = (size_t)n * a.Cin * plane;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int ci = c0 + (s_lds[j] >> 20);
        const bool ok = (s_lds[j] >= 0) && (s_goff[j] >= 0) && (ci < a.Cin);
        rin[j] = ok ? IO::ld4(a.in, in_n + (size_t)ci * plane + s_goff[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      have_w = load_w;
      if (load_w) {
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
          const int idx = tid + j * 256;
          rw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx < G::W_ITEMS) {
            const int j4 = idx & 3, row = idx >> 2;          // row = q*CK + c
            const int c = row % CK, q = row / CK;
            if (c0 + c < a.cin_pad) {
              // the product's weight: the sum of the 3x3 taps that land on this stored pixel (MODE 0), or the one tap of the data-gradient (MODE 1);
              // fixed order (ky outer, kx inner), fp32 adds
              SubCombo cb_{};
#pragma unroll
              for (int qq = 0; qq < NCOMBO; ++qq) if (qq == q) cb_ = sub_combo<MODE>(qq);
              float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
              bool first = true;
              for (int ky = cb_.ky0; ky <= cb_.ky1; ++ky)
                for (int kx = cb_.kx0; kx <= cb_.kx1; ++kx) {
                  const float4 w4 = *reinterpret_cast<const float4*>(a.w + ((size_t)(ky * 3 + kx) * a.cin_pad + c0 + c) * a.cout_pad + co0 + j4 * 4);
                  if (first) { acc4 = w4; first = false; }
                  else { acc4.x += w4.x; acc4.y += w4.y; acc4.z += w4.z; acc4.w += w4.w; }
                }
              rw[j] = acc4;
            }
          }
        }
      }
    };
    auto store_chunk = [&](float* buf) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (s_lds[j] < 0) continue;
        *reinterpret_cast<float4*>(buf + (s_lds[j] & 0xFFFFF)) = rin[j];
      }
      if (have_w) {
        float* w_lds = buf + CK * PS;
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
          const int idx = tid + j * 256;
          if (idx < G::W_ITEMS) *reinterpret_cast<float4*>(w_lds + (idx >> 2) * WS + (idx & 3) * 4) = rw[j];
        }
      }
    };
    int key_cb[2] = {-1, -1}, key_c0[2] = {-1, -1};
    int item = vb, chunk = 0, n, tile, cb, tile_set = -1;
    decode(item, n, tile, cb);
    set_tile(tile); tile_set = tile;
    load_chunk(n, cb * 16, 0, true);
    lds_barrier();                                    // barrier #0
    for (int p = 0; p < T; ++p) {
      store_chunk(smem + (p & 1) * BUF);
      key_cb[p & 1] = cb; key_c0[p & 1] = chunk * CK;
      if (p + 1 < T) {
        if (++chunk == nchunks) { chunk = 0; item += gridDim.x; decode(item, n, tile, cb); }
        if (tile != tile_set) { set_tile(tile); tile_set = tile; }
        const int b = (p + 1) & 1;
        load_chunk(n, cb * 16, chunk * CK, !(key_cb[b] == cb && key_c0[b] == chunk * CK));
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
  // M-tile i: stored row 2*wave + (i >> 1), stored columns (i & 1)*16 .. +15; LDS element (channel k, row + 1 + sy, column + 4 + sx)
  const int a_lane = k * PS + (2 * wave + 1) * RS + 4 + m;
  const int b_lane = CK * PS + k * WS + m;
  auto mt_off = [](int i) { return (i >> 1) * RS + (i & 1) * 16; };

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
        // MODE 1 needs 4 of the 9 shifts: its A fragments are double buffered (the next M-tile's are requested before this one's MFMAs).
        // MODE 0 needs all 9 and 16 B fragments: one set (the partner wave of the SIMD covers the LDS latency; a second set spills).
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
  auto epilogue = [&](int n, int tile, int co0) {
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int y0 = ty * TLH + 2 * wave, x0 = tx * TLW + 4 * k;       // stored coordinates of this lane's first pixel quad (M-tile 0)
    const int co = co0 + m;
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
      for (int i = 0; i < 4; ++i) {
        const int y = y0 + (i >> 1), x = x0 + (i & 1) * 16;
        if (y < a.Hs && x < a.Ws) cnt += 16.f;
      }
      if (cnt > 0.f) {
        const float rc = __builtin_amdgcn_rcpf(cnt);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int y = y0 + (i >> 1), x = x0 + (i & 1) * 16;
          const bool ok = (y < a.Hs && x < a.Ws);
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += ok ? acc[i][p][r] : 0.f;
        }
        const float mean = s * rc;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int y = y0 + (i >> 1), x = x0 + (i & 1) * 16;
          const bool ok = (y < a.Hs && x < a.Ws);
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = acc[i][p][r] - mean; qq += ok ? d * d : 0.f; }
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
      const size_t pb = ((size_t)n * a.Cout + co) * a.Hout * Wo;               // element offset of this lane's output plane (out, mk_u, mk_ref share the shape)
      // mask reference: the materialised activation (mk_ref), or - when the activation was never written (mk_ref == NULL: lrelu(bn(u)) without a
      // residual add) - its pre-activation sc*u + sh recomputed from u
      const bool have_ref = (a.epi_mode == 3 && mk_ref != nullptr);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int y = y0 + (i >> 1), x = x0 + (i & 1) * 16;
        if (y >= a.Hs || x >= a.Ws) continue;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const size_t off = pb + (size_t)(2 * y + py) * Wo + 2 * x;
          const f32x4 e = acc[i][py * 2], o = acc[i][py * 2 + 1];
          float4 v0 = make_float4(e[0], o[0], e[1], o[1]), v1 = make_float4(e[2], o[2], e[3], o[3]);
          if (a.epi_mode == 3) {
            const float4 u0 = IO::ld4(a.mk_u, off), u1 = IO::ld4(a.mk_u, off + 4);
            float4 r0, r1;
            if (have_ref) { r0 = IO::ld4(mk_ref, off); r1 = IO::ld4(mk_ref, off + 4); }
            else {
              r0 = make_float4(mk_sc * u0.x + mk_sh, mk_sc * u0.y + mk_sh, mk_sc * u0.z + mk_sh, mk_sc * u0.w + mk_sh);
              r1 = make_float4(mk_sc * u1.x + mk_sh, mk_sc * u1.y + mk_sh, mk_sc * u1.z + mk_sh, mk_sc * u1.w + mk_sh);
            }
            v0.x *= (r0.x > 0.f) ? 1.f : a.mk_slope; v0.y *= (r0.y > 0.f) ? 1.f : a.mk_slope; v0.z *= (r0.z > 0.f) ? 1.f : a.mk_slope; v0.w *= (r0.w > 0.f) ? 1.f : a.mk_slope;
            v1.x *= (r1.x > 0.f) ? 1.f : a.mk_slope; v1.y *= (r1.y > 0.f) ? 1.f : a.mk_slope; v1.z *= (r1.z > 0.f) ? 1.f : a.mk_slope; v1.w *= (r1.w > 0.f) ? 1.f : a.mk_slope;
            s1 += ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w));
            s2 += ((v0.x * (u0.x - mk_mu) + v0.y * (u0.y - mk_mu)) + (v0.z * (u0.z - mk_mu) + v0.w * (u0.w - mk_mu))) +
                  ((v1.x * (u1.x - mk_mu) + v1.y * (u1.y - mk_mu)) + (v1.z * (u1.z - mk_mu) + v1.w * (u1.w - mk_mu)));
          }
          IO::st4(a.out, off, v0);
          IO::st4(a.out, off + 4, v1);
        }
      }
      if (a.epi_mode == 3) { st_mean[0] += s1; st_m2[0] += s2; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  int item = vb, chunk = 0, n, tile, cb;
  decode(item, n, tile, cb);
  load_bias(cb * 16);
  lds_barrier();                                      // barrier #0
  lds_barrier();                                      // barrier #1: chunk 0 is in buffer 0
  for (int p = 0; p < T; ++p) {
    const int c0 = chunk * CK;
    const int ncg = min(CK / 4, (a.cin_pad - c0) / 4);
    compute(smem + (p & 1) * BUF, ncg);
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

template <int MODE, typename AT>
int launch_conv_subpix_t(ConvArgs a, const float* mk_ref, hipStream_t st) {
  using G = SubGeo<MODE>;
  const size_t lds_bytes = sizeof(float) * 2 * (size_t)G::BUF;
  static std::once_flag attr_once;
  std::call_once(attr_once, []() { (void)hipFuncSetAttribute((const void*)conv_subpix_kernel<MODE, AT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)); });
  a.tiles_x = cdiv(a.Ws, G::TLW); a.tiles_y = cdiv(a.Hs, G::TLH);
  a.ncb = cdiv(a.Cout, 16);
  const long nitems = (long)a.N * a.tiles_x * a.tiles_y * a.ncb;
  const int per_cu = std::max(1, std::min(2, (int)((160 * 1024) / (lds_bytes + 256))));
  long nblocks = std::min<long>(nitems, (long)num_cus() * per_cu);
  if (nblocks > a.ncb) nblocks -= nblocks % a.ncb;
  MS_LAUNCH((conv_subpix_kernel<MODE, AT>), dim3((unsigned)nblocks), dim3(512), lds_bytes, st, a, mk_ref);
  return check_launch("conv_subpix");
}
template <int MODE>
int launch_conv_subpix(const ConvArgs& a, const float* mk_ref, hipStream_t st) {
  return a.act_bf16 ? launch_conv_subpix_t<MODE, ms_bf16>(a, mk_ref, st) : launch_conv_subpix_t<MODE, float>(a, mk_ref, st);
}

}  // namespace ms
