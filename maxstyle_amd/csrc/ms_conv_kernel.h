// Implicit-GEMM convolution kernel template for gfx950 (exact-fp32 matrix cores, v_mfma_f32_16x16x4_f32).
// See ms_conv.hip for what it replaces in the reference and for the dispatch; this header is included by the
// translation units that instantiate slices of the template (parallel compilation).
//
// Workgroup = 4 waves = one output tile (8x32 pixels, or 16x16 for small feature maps) x 16*NT output channels.
// K loop over chunks of CK input channels:
//     global --(16-B loads of an aligned window, issued back to back)--> registers   [prefetch of chunk i+1]
//     registers --(BatchNorm apply / BatchNorm backward prologue)--> LDS               [chunk i]
//     LDS --ds_read_b32 fragments--> MFMA 16x16x4 (A = 16 pixels x 4 channels, B = 4 channels x 16 couts)
// The loads of chunk i+1 are in flight while chunk i is multiplied, so a workgroup hides its own HBM/L2 latency
// even at one workgroup per CU (deep layers: few tiles, long K).
#pragma once
#include <algorithm>
#include "ms_common.h"

namespace ms {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kNumCU = 256;      // MI355X: 8 XCDs x 32 CUs

enum { FETCH_NORMAL = 0, FETCH_UPS2 = 1, FETCH_ZINS2 = 2 };

struct ConvArgs {
  const float* in; const float* in2; float* out; const float* w; const float* bias;
  const float* pro_a; const float* pro_b; const float* pro_c;
  float* stats;                 // float4 [cout][nparts] (count, mean, M2, 0) or null
  int N, Cin, Hs, Ws, Hin, Win, Cout, Hout, Wout, cin_pad, cout_pad;
  int pro_mode, pro_nstride, pro_cstride; float slope;
  int epi_mode, tiles_x, tiles_y, cout_real, ncb;   // ncb: number of output-channel blocks (of 16*NT)
};

template <int KS, int STRIDE, bool VEC, bool NARROW>
struct Geo {
  static constexpr int TW = NARROW ? 16 : 32;
  static constexpr int TH = NARROW ? 16 : 8;
  static constexpr int PAD = (KS == 3) ? 1 : 0;
  static constexpr int PADL = VEC ? ((KS == 3) ? 4 : 0) : PAD;           // window starts PADL logical columns left of ox0*S
  static constexpr int IH = (TH - 1) * STRIDE + KS;
  static constexpr int WIN_W = VEC ? (TW * STRIDE + ((KS == 3) ? 8 : 0)) : ((TW - 1) * STRIDE + KS);
  static constexpr int HALF = (WIN_W + 1) / 2;
  static constexpr int RS = (STRIDE == 1) ? ((WIN_W + 3) / 4 * 4) : 2 * HALF;
  static constexpr int BASE = IH * RS;
  static constexpr int PS = BASE + ((16 - BASE % 32 + 32) % 32);           // plane stride == 16 (mod 32 banks)
  static constexpr int CK = (STRIDE == 1 && VEC) ? 16 : 8;   // scalar fallback stages 4x more slots per channel: halve the chunk
  static constexpr int VW = VEC ? 4 : 1;
  static constexpr int ROW_ITEMS = WIN_W / VW;                              // VEC: WIN_W % 4 == 0 by construction
  static constexpr int ITEMS = CK * IH * ROW_ITEMS;
  static constexpr int NI = (ITEMS + 255) / 256;
  // LDS column of logical tap offset t = kx - PAD + PADL (added to the pixel's x inside the tile)
  static constexpr int tap_col(int kx) {
    const int t = kx - PAD + PADL;
    return (STRIDE == 1) ? t : ((t & 1) * HALF + (t >> 1));
  }
};

template <int NT> struct WGeo { static constexpr int WS = (NT == 1) ? 16 : NT * 16 + 16; };

// waves per SIMD the register budget is sized for (a workgroup is 4 waves = 1 per SIMD): 4 -> <=128 VGPRs, 3 -> <=168, 2 -> <=256
template <int NT, bool IN2, int STRIDE> struct Occ { static constexpr int W = (NT == 1 && !IN2 && STRIDE == 1) ? 3 : 2; };

// Persistent workgroups: work item = (image n, output tile, output-channel block); a workgroup walks items
// blockIdx.x, blockIdx.x+gridDim.x, ... and software-pipelines over the flattened (item, K-chunk) sequence, so the
// global loads of the NEXT tile/chunk are in flight while the current one is multiplied - the matrix pipe stays fed even
// when all workgroups of a CU run in lock-step (measured: without this the MFMA pipe was 35 % busy, waves 55 % issue-stalled).
template <int KS, int STRIDE, int FETCH, int NT, bool VEC, bool NARROW, bool IN2>
__global__ __launch_bounds__(256, (Occ<NT, IN2, STRIDE>::W)) void conv_mfma_kernel(const ConvArgs a) {
  using G = Geo<KS, STRIDE, VEC, NARROW>;
  constexpr int CK = G::CK, PS = G::PS, RS = G::RS, IH = G::IH, PAD = G::PAD, PADL = G::PADL, HALF = G::HALF;
  constexpr int TW = G::TW, TH = G::TH, VW = G::VW, ROW_ITEMS = G::ROW_ITEMS, ITEMS = G::ITEMS, NI = G::NI;
  constexpr int WS = WGeo<NT>::WS;
  constexpr int TAPS = KS * KS;
  constexpr int COUT_TILE = 16 * NT;
  constexpr int WITEMS = TAPS * CK * (COUT_TILE / 4);
  constexpr int NWI = (WITEMS + 255) / 256;
  static_assert(!VEC || FETCH == FETCH_NORMAL, "vector staging needs the plain fetch");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* in_lds = smem;                 // [CK][PS]
  float* w_lds = smem + CK * PS;        // [TAPS][CK][WS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, k = lane >> 4;
  const int ntiles = a.tiles_x * a.tiles_y;
  const int ncb = a.ncb;
  const int nitems = a.N * ntiles * ncb;
  const int nchunks = (a.cin_pad + CK - 1) / CK;
  const size_t in_plane = (size_t)a.Hs * a.Ws;

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- per-thread staging slots of the item being LOADED: slot -> (channel c, row r, window column w) ----
  int s_lds[NI];        // (channel-in-chunk << 20) | LDS float offset of the slot, or -1: no slot   (tile independent)
  int s_goff[NI];       // global offset inside one channel plane (or -1: out of the image -> zeros)  (per tile)
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int item = tid + j * 256;
    s_lds[j] = -1;
    if (item < ITEMS) {
      const int f = item % ROW_ITEMS;
      const int row = item / ROW_ITEMS;
      const int r = row % IH, c = row / IH;
      const int w = f * VW;
      const int q = (STRIDE == 1) ? w : ((w & 1) * HALF + (w >> 1));
      s_lds[j] = (c << 20) | (c * PS + r * RS + q);
    }
  }
  auto set_tile = [&](int tile) {          // recompute the global offsets of the slots for a new output tile
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int iy0 = ty * TH * STRIDE - PAD;
    const int wx0 = tx * TW * STRIDE - PADL;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int item = tid + j * 256;
      const int f = item % ROW_ITEMS;
      const int r = (item / ROW_ITEMS) % IH;
      const int Y = iy0 + r, X = wx0 + f * VW;
      bool ok = (item < ITEMS) && (Y >= 0) && (Y < a.Hin) && (X >= 0) && (X < a.Win);
      int ys = Y, xs = X;
      if (FETCH == FETCH_UPS2) { ys = Y >> 1; xs = X >> 1; }
      if (FETCH == FETCH_ZINS2) { ok = ok && !((Y | X) & 1); ys = Y >> 1; xs = X >> 1; }
      s_goff[j] = ok ? (ys * a.Ws + xs) : -1;
    }
  };

  float rin[NI][VW];
  float rin2[IN2 ? NI : 1][VW];
  float4 rw[NWI];

  auto load_chunk = [&](int n, int co0, int c0, bool load_w) {
    const float* in_n = a.in + (size_t)n * a.Cin * in_plane;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int ci = c0 + (s_lds[j] >> 20);
      const bool ok = (s_lds[j] >= 0) && (s_goff[j] >= 0) && (ci < a.Cin);
      const size_t off = ok ? ((size_t)ci * in_plane + (size_t)s_goff[j]) : 0;
      if constexpr (VEC) {
        const float4 v = ok ? *reinterpret_cast<const float4*>(in_n + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        rin[j][0] = v.x; rin[j][1] = v.y; rin[j][2] = v.z; rin[j][3] = v.w;
        if constexpr (IN2) {
          const float* in2_n = a.in2 + (size_t)n * a.Cin * in_plane;
          const float4 u = ok ? *reinterpret_cast<const float4*>(in2_n + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          rin2[j][0] = u.x; rin2[j][1] = u.y; rin2[j][2] = u.z; rin2[j][3] = u.w;
        }
      } else {
        rin[j][0] = ok ? in_n[off] : 0.f;
        if constexpr (IN2) { const float* in2_n = a.in2 + (size_t)n * a.Cin * in_plane; rin2[j][0] = ok ? in2_n[off] : 0.f; }
      }
    }
    if (load_w) {
#pragma unroll
      for (int j = 0; j < NWI; ++j) {
        const int idx = tid + j * 256;
        const int j4 = idx % (COUT_TILE / 4);
        const int row = idx / (COUT_TILE / 4);      // tap*CK + c
        const int c = row % CK, tap = row / CK;
        rw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < WITEMS && c0 + c < a.cin_pad)
          rw[j] = *reinterpret_cast<const float4*>(a.w + ((size_t)tap * a.cin_pad + c0 + c) * a.cout_pad + co0 + j4 * 4);
      }
    }
  };

  // s_ok bit j: slot j of the chunk held in registers was inside the image (its prologue must be applied)
  auto store_chunk = [&](int n, int c0, unsigned okmask, bool store_w) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (s_lds[j] < 0) continue;
      const int ci = c0 + (s_lds[j] >> 20);
      const bool ok = ((okmask >> j) & 1u) && (ci < a.Cin);
      float v[VW];
#pragma unroll
      for (int e = 0; e < VW; ++e) v[e] = rin[j][e];
      if (ok) {
        if (a.pro_mode == 1) {
          const int pi = (n * a.pro_nstride + ci) * a.pro_cstride;
          const float pa = a.pro_a[pi], pb = a.pro_b[pi];
#pragma unroll
          for (int e = 0; e < VW; ++e) v[e] = leaky(pa * v[e] + pb, a.slope);
        } else if constexpr (IN2) {
          if (a.pro_mode == 2) {
            const int pi = ci * a.pro_cstride;
            const float pa = a.pro_a[pi], pb = a.pro_b[pi], pc = a.pro_c[pi];
#pragma unroll
            for (int e = 0; e < VW; ++e) v[e] = pa * v[e] + pb * rin2[j][e] + pc;
          }
        }
      }
      float* dst = in_lds + (s_lds[j] & 0xFFFFF);
      if constexpr (VEC) {
        if constexpr (STRIDE == 1) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {          // de-interleave even / odd columns (w % 4 == 0 -> even half at dst, odd half at dst+HALF)
          *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[2]);
          *reinterpret_cast<float2*>(dst + HALF) = make_float2(v[1], v[3]);
        }
      } else {
        dst[0] = v[0];
      }
    }
    if (store_w) {
#pragma unroll
      for (int j = 0; j < NWI; ++j) {
        const int idx = tid + j * 256;
        if (idx < WITEMS) {
          const int j4 = idx % (COUT_TILE / 4);
          const int row = idx / (COUT_TILE / 4);
          *reinterpret_cast<float4*>(w_lds + row * WS + j4 * 4) = rw[j];
        }
      }
    }
  };
  auto okmask_now = [&]() { unsigned mk = 0; 
#pragma unroll
    for (int j = 0; j < NI; ++j) mk |= (s_goff[j] >= 0 ? 1u : 0u) << j;
    return mk; };

  // M-tile i of this wave: NARROW: rows 4*wave+i, columns 0..15; else rows 2*wave+(i>>1), columns (i&1)*16..
  const int a_lane = k * PS + m;
  const int b_lane = k * WS + m;
  auto mt_row = [&](int i) { return NARROW ? (wave * 4 + i) : (wave * 2 + (i >> 1)); };
  auto mt_col = [&](int i) { return NARROW ? 0 : ((i & 1) * 16); };

  auto compute = [&](int ncg) {
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
      const int ky = tap / KS, kx = tap - ky * KS;
      const int t = kx - PAD + PADL;
      const int tap_off = ky * RS + ((STRIDE == 1) ? t : ((t & 1) * HALF + (t >> 1)));
      const float* ap = in_lds + a_lane + tap_off;
      const float* bp = w_lds + b_lane + tap * CK * WS;
#pragma unroll
      for (int cg = 0; cg < CK / 4; ++cg) {
        if (cg < ncg) {
          float bf[NT], af[4];
#pragma unroll
          for (int j = 0; j < NT; ++j) bf[j] = bp[cg * 4 * WS + j * 16];
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i] = ap[cg * 4 * PS + mt_row(i) * STRIDE * RS + mt_col(i)];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  // ---- epilogue of one finished item (registers only + global stores; no LDS, no barriers) ----
  auto epilogue = [&](int n, int tile, int co0) {
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int xq = 4 * k;     // D layout (16x16): column (output channel) = lane&15, rows (pixels) = 4*(lane>>4) + reg
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = co0 + j * 16 + m;
      int bidx = co;
      if (a.epi_mode == 2) bidx = co % a.cout_real;
      const float bv = (a.bias != nullptr && co < a.Cout) ? a.bias[bidx] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] += bv;
    }
    if (a.stats != nullptr) {
      // per-WAVE (count, mean, M2) of this wave's pixels for each output channel: two passes over registers,
      // cross-lane combine of the 4 lane groups that share a channel (xor 16, 32). Partials: [co][(n*ntiles+tile)*4 + wave]
      int rows_ok = 0, cols_ok[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { cols_ok[i] = 0; }
      float cnt = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int y = oy0 + mt_row(i);
        const int x0 = ox0 + mt_col(i);
        const int nx = min(16, a.Wout - x0);
        if (y < a.Hout && nx > 0) cnt += (float)nx;
      }
      (void)rows_ok; (void)cols_ok;
      const int nparts = a.N * ntiles * 4;
      const int pidx = (n * ntiles + tile) * 4 + wave;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int y = oy0 + mt_row(i);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int x = ox0 + mt_col(i) + xq + r;
            s += ((y < a.Hout) && (x < a.Wout)) ? acc[i][j][r] : 0.f;
          }
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mean = cnt > 0.f ? s / cnt : 0.f;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int y = oy0 + mt_row(i);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int x = ox0 + mt_col(i) + xq + r;
            const float d = acc[i][j][r] - mean;
            q += ((y < a.Hout) && (x < a.Wout)) ? d * d : 0.f;
          }
        }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const int co = co0 + j * 16 + m;
        if (k == 0 && co < a.Cout) reinterpret_cast<float4*>(a.stats)[(size_t)co * nparts + pidx] = make_float4(cnt, mean, q, 0.f);
      }
    }
    if (a.epi_mode == 2) {
      // ConvTranspose2d k=2 s=2: GEMM column j = (dy*2+dx)*cout_real + co -> out[n,co,2y+dy,2x+dx]
      const int Ho = 2 * a.Hout, Wo = 2 * a.Wout;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = co0 + j * 16 + m;
        if (col >= 4 * a.cout_real) continue;
        const int q = col / a.cout_real, co = col - q * a.cout_real;
        const int dy = q >> 1, dx = q & 1;
        float* op = a.out + ((size_t)n * a.cout_real + co) * Ho * Wo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int y = oy0 + mt_row(i);
          if (y >= a.Hout) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int x = ox0 + mt_col(i) + xq + r;
            if (x < a.Wout) op[(size_t)(2 * y + dy) * Wo + 2 * x + dx] = acc[i][j][r];
          }
        }
      }
    } else {
      const bool vec_ok = (a.Wout % 4 == 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = co0 + j * 16 + m;
        if (co >= a.Cout) continue;
        float* op = a.out + ((size_t)n * a.Cout + co) * a.Hout * a.Wout;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int y = oy0 + mt_row(i);
          if (y >= a.Hout) continue;
          const int x = ox0 + mt_col(i) + xq;
          float* o = op + (size_t)y * a.Wout + x;
          if (vec_ok && x + 3 < a.Wout) {
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (a.epi_mode == 1) { const float4 p = *reinterpret_cast<const float4*>(o); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
            *reinterpret_cast<float4*>(o) = v;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (x + r < a.Wout) o[r] = (a.epi_mode == 1) ? (o[r] + acc[i][j][r]) : acc[i][j][r];
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- pipelined walk over (item, chunk) ----
  int item = blockIdx.x;
  if (item >= nitems) return;
  auto decode = [&](int it, int& n, int& tile, int& cb) { cb = it % ncb; const int t2 = it / ncb; tile = t2 % ntiles; n = t2 / ntiles; };
  int n_cur, tile_cur, cb_cur;
  decode(item, n_cur, tile_cur, cb_cur);
  set_tile(tile_cur);
  unsigned ok_ld = okmask_now();
  load_chunk(n_cur, cb_cur * COUT_TILE, 0, true);
  int lds_w_cb = -1, lds_w_c0 = -1;      // which weight slice w_lds currently holds
  int chunk = 0;
  while (true) {
    const int c0 = chunk * CK;
    const bool w_fresh = !(lds_w_cb == cb_cur && lds_w_c0 == c0);
    __syncthreads();                       // every wave finished multiplying the previous chunk
    store_chunk(n_cur, c0, ok_ld, w_fresh);
    lds_w_cb = cb_cur; lds_w_c0 = c0;
    __syncthreads();
    // what comes next: another chunk of this item, or the first chunk of this workgroup's next item
    int n_nx = n_cur, tile_nx = tile_cur, cb_nx = cb_cur, chunk_nx = chunk + 1, item_nx = item;
    if (chunk_nx == nchunks) { chunk_nx = 0; item_nx = item + gridDim.x; if (item_nx < nitems) decode(item_nx, n_nx, tile_nx, cb_nx); }
    const bool have_next = item_nx < nitems;
    if (have_next) {
      if (tile_nx != tile_cur) set_tile(tile_nx);
      ok_ld = okmask_now();
      const bool w_needed = !(cb_nx == cb_cur && chunk_nx * CK == c0);
      load_chunk(n_nx, cb_nx * COUT_TILE, chunk_nx * CK, w_needed);      // in flight while this chunk is multiplied
    }
    const int ncg = min(CK / 4, (a.cin_pad - c0) / 4);
    if (ncg == CK / 4) compute(CK / 4); else compute(ncg);
    if (chunk + 1 == nchunks) epilogue(n_cur, tile_cur, cb_cur * COUT_TILE);
    if (!have_next) break;
    item = item_nx; n_cur = n_nx; tile_cur = tile_nx; cb_cur = cb_nx; chunk = chunk_nx;
  }
}

template <int KS, int STRIDE, int FETCH, int NT, bool VEC, bool NARROW, bool IN2>
int launch_conv(const ConvArgs& a, hipStream_t st) {
  using G = Geo<KS, STRIDE, VEC, NARROW>;
  constexpr int lds_floats = G::CK * G::PS + KS * KS * G::CK * WGeo<NT>::WS;
  constexpr size_t lds_bytes = sizeof(float) * lds_floats;
  static bool attr_set = false;
  if (!attr_set && lds_bytes > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_mfma_kernel<KS, STRIDE, FETCH, NT, VEC, NARROW, IN2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_set = true;
  }
  const long nitems = (long)a.N * a.tiles_x * a.tiles_y * a.ncb;
  // resident workgroups per CU: limited by the register budget (Occ) and LDS (160 KiB per CU)
  const int per_cu = std::max(1, std::min(Occ<NT, IN2, STRIDE>::W, (int)((160 * 1024) / (lds_bytes + 512))));
  const long nblocks = std::min<long>(nitems, (long)kNumCU * per_cu);
  dim3 grid((unsigned)nblocks), block(256);
  hipLaunchKernelGGL((conv_mfma_kernel<KS, STRIDE, FETCH, NT, VEC, NARROW, IN2>), grid, block, lds_bytes, st, a);
  return check_launch("conv_mfma");
}

// dispatch entry points implemented in the ms_conv_inst*.hip translation units
int conv_dispatch_k3s1(const ConvArgs& a, int fetch, int nt, bool vec, bool narrow, bool in2, hipStream_t st);
int conv_dispatch_k1s1(const ConvArgs& a, int nt, bool vec, bool narrow, bool in2, hipStream_t st);
int conv_dispatch_s2(const ConvArgs& a, int ks, int nt, bool vec, bool narrow, hipStream_t st);

}  // namespace ms
