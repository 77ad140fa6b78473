// LDS-tiled GEMM form of the 1x1 convolutions with MANY input channels on SMALL images (config 4: 256 .. 512 channels on 20 x 20 / 40 x 40 pixels - the skip paths of
// the deep residual blocks and their data-gradients, encoder_decoder.py:62-64, 344-346).  The tiled first-generation kernel runs them with 16-channel output tiles
// (its 64-channel tile needs 205 registers: one workgroup per CU) on 8 x 32-pixel tiles that a 20 x 20 image fills to 52 %: 0.16-0.26 of the MFMA bound
// (profiles/r04_experiments.txt 13).  The streaming kernel (ms_conv_k1s.h) cannot hold their weight slice in LDS.  This one is the stride-2 / sub-pixel second generation's
// recipe for a conv without a halo:
//   * a work item = 4 units of 64 consecutive pixels (one per MFMA wave) of a flattened (image, unit) list x NT sixteen-channel output blocks: a 20 x 20 image is 6.25 units;
//   * per 8-channel chunk the four units' rows and the weight slice travel to LDS by LDS-DMA (three stage buffers, nothing through registers);
//   * the MFMA waves read A fragments (16 pixels x 4 channels) and B fragments with 32-bit LDS reads on disjoint banks (row strides 80 / 16 NT + 16 floats).
// Channels are accumulated in ascending 4-channel groups, like every 1x1 kernel of the library: same bits.  Epilogues: plain (+ bias) and the residual tail
// lrelu((sc u + sh) + (acc + bias)); carries the rider and the cross-workgroup finalize (epilogue kind) like the kernels it replaces.  fp32 storage, no prologue.
#pragma once
#include <cstdlib>
#include <map>
#include <mutex>
#include "ms_conv_kernel.h"

namespace ms {

template <int NT>
struct K1gGeo {
  static constexpr int CK = 8;
  static constexpr int WSW = (NT == 1) ? 16 : 16 * NT + 16;
  static constexpr int WROWP = WSW / 4;
  static constexpr int W_ITEMS = CK * WROWP;
  static constexpr int NWJ = ((W_ITEMS + 63) / 64 + 3) / 4;            // 1
  static constexpr int W_FLOATS = 4 * NWJ * 256;
  static constexpr int AROW = 80;                                      // 64 pixels + 16 floats of padding: the four k-rows of an A fragment on disjoint banks
  static constexpr int AROWP = AROW / 4;
  static constexpr int IN_ITEMS = 4 * CK * AROWP;                       // 640 pieces
  static constexpr int NJ = ((IN_ITEMS + 63) / 64 + 3) / 4;             // 3
  static constexpr int IN_FLOATS = 4 * NJ * 256;
  static constexpr int BUF = IN_FLOATS + W_FLOATS;
  static constexpr int KDMA = NJ + NWJ;
  static constexpr int OOB = (int)0x80000000;
};

template <int NT, int EPI>          // EPI 0 plain | 4 residual tail | 5 residual tail at twice the resolution (every value feeds a 2 x 2 block of outputs)
__global__ __launch_bounds__(512, 4) void conv_k1g_kernel(const ConvArgs a) {
  using G = K1gGeo<NT>;
  constexpr int CK = G::CK, WSW = G::WSW, AROW = G::AROW, BUF = G::BUF, NJ = G::NJ, NWJ = G::NWJ, OOB = G::OOB;
  constexpr int COUT_TILE = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = MS_TID >> 6, lane = MS_TID & 63;
  const bool producer = wave >= 4;
  const int ncb = a.ncb;
  const int HW = a.Hs * a.Ws;
  const int upi = (HW + 63) >> 6, U = a.N * upi;
  const int ngroups = (U + 3) >> 2;
  const int nitems = ngroups * ncb;
  const int nchunks = a.cin_pad / CK;
  const int vb = ((int)gridDim.x % 8 == 0) ? (((int)blockIdx.x % 8) * ((int)gridDim.x / 8) + (int)blockIdx.x / 8) : (int)blockIdx.x;
  const int my_items = (nitems - vb + (int)gridDim.x - 1) / (int)gridDim.x;
  const int T = my_items * nchunks;
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto decode = [&](int it, int& grp, int& cb) { cb = it % ncb; grp = it / ncb; };

  if (producer) {
    // =========================================== STAGING waves: LDS-DMA only ===========================================
    __builtin_amdgcn_s_setprio(3);
    const int sw = __builtin_amdgcn_readfirstlane(wave) - 4;
    const ms_i32x4 rs_in = ms_dma_rsrc_n(a.in, (unsigned)a.N * a.Cin * HW * 4u);
    const ms_i32x4 rs_w = ms_dma_rsrc_n(a.w, (unsigned)a.cin_pad * a.cout_pad * 4u);
    const unsigned lds0 = ms_lds_addr(smem);
    int w_voff[NWJ];
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
      const int idx = (sw + 4 * j) * 64 + lane;
      const int c = idx / G::WROWP, p = idx - c * G::WROWP;
      w_voff[j] = (idx < G::W_ITEMS && p < 4 * NT) ? (int)(((size_t)c * a.cout_pad + 4 * p) * 4) : OOB;
    }
    int i_pk[NJ], i_voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int idx = (sw + 4 * j) * 64 + lane;
      const int wv = idx / (CK * G::AROWP), rem = idx - wv * (CK * G::AROWP), c = rem / G::AROWP, p = rem - c * G::AROWP;
      i_pk[j] = (idx < G::IN_ITEMS && p < 16) ? ((wv << 16) | (c << 8) | p) : -1;
      i_voff[j] = OOB;
    }
    auto set_group = [&](int grp) {
      // lanes 0..3: image and first pixel of the group's four units
      const int u = grp * 4 + (lane & 3);
      const int n = u / upi, p0 = (u - n * upi) << 6;
      const int base = (u < U) ? (n * a.Cin * HW + p0) : OOB;
      const int lim = (u < U) ? (HW - p0) : 0;                          // pixels of the unit inside the image
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int pk = i_pk[j];
        const int wv = (pk >> 16) & 3;
        const int bb = __shfl(base, wv, 64), ll = __shfl(lim, wv, 64);
        const int c = (pk >> 8) & 0xFF, p = pk & 0xFF;
        i_voff[j] = (pk >= 0 && bb != OOB && 4 * p < ll) ? ((bb + c * HW + 4 * p) * 4) : OOB;
      }
    };
    auto issue = [&](int buf, int cb, int chunk) {
      const unsigned lb = lds0 + (unsigned)buf * (BUF * 4);
      const int c0 = chunk * CK;
#pragma unroll
      for (int j = 0; j < NWJ; ++j) ms_lds_dma16(rs_w, lb + G::IN_FLOATS * 4 + (unsigned)(sw + 4 * j) * 1024, w_voff[j], (c0 * a.cout_pad + cb * COUT_TILE) * 4);
#pragma unroll
      for (int j = 0; j < NJ; ++j) ms_lds_dma16(rs_in, lb + (unsigned)(sw + 4 * j) * 1024, i_voff[j], c0 * HW * 4);
    };
    int item = vb, chunk = 0, grp, cb, grp_set = -1, ring = 0;
    decode(item, grp, cb);
    auto issue_next = [&](bool more) {
      if (grp != grp_set) { set_group(grp); grp_set = grp; }
      issue(ring, cb, chunk);
      if (++ring == 3) ring = 0;
      if (++chunk == nchunks) { chunk = 0; item += gridDim.x; if (more) decode(item, grp, cb); }
    };
    lds_barrier();                                    // barrier #0
    issue_next(T > 1);
    for (int p = 0; p < T; ++p) {
      if (p + 1 < T) { issue_next(p + 2 < T); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::KDMA) : "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();                                  // barrier #(p+1)
    }
    return;
  }

  // =========================================== MFMA waves ===========================================
  const int m = lane & 15, k = lane >> 4;
  unsigned xf_tag = 0u;
  int xf_nparts = 0;
  const bool xf_epi = (EPI != 0) && (a.xf_tab != nullptr);

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int a_lane = (wave * CK + k) * AROW + m;
  const int b_lane = G::IN_FLOATS + k * WSW + m;
  auto compute = [&](const float* buf) {
#pragma unroll
    for (int cg = 0; cg < CK / 4; ++cg) {
      float bf[NT], af[4];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = buf[b_lane + cg * 4 * WSW + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = buf[a_lane + cg * 4 * AROW + 16 * i];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  float bias_v[NT], mk_sc[NT], mk_sh[NT];
  int cur_cb = -1;
  bool xf_pending = xf_epi;
  auto load_cb = [&](int cb) {
    if (cb == cur_cb) return;
    cur_cb = cb;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = cb * COUT_TILE + j * 16 + m;
      bias_v[j] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
      mk_sc[j] = mk_sh[j] = 0.f;
      if (EPI != 0 && !xf_epi) {
        const float4 cf = (co < a.Cout) ? reinterpret_cast<const float4*>(a.mk_coef)[co] : make_float4(0.f, 0.f, 0.f, 0.f);
        mk_sc[j] = cf.x; mk_sh[j] = cf.y;
      }
    }
  };
  // D layout: this lane holds output channel m of pixels 16 i + 4 k .. + 3 of its wave's unit
  auto epilogue = [&](int grp, int cb) {
    if (xf_pending) {              // (a workgroup keeps one channel block for its life: grid % ncb == 0)
      xf_pending = false;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int co = cb * COUT_TILE + j * 16 + m;
        conv_u64_t g[2];
        xfin_peek(a, min(co, a.xf_C - 1), vb & (kXfinRep - 1), g);
        const float2 cf = (co < a.Cout) ? xfin_poll(a, co, vb & (kXfinRep - 1), xf_tag, g) : make_float2(0.f, 0.f);
        mk_sc[j] = cf.x; mk_sh[j] = cf.y;
      }
    }
    const int u = grp * 4 + wave;
    const int n = u / upi, p0 = (u - n * upi) << 6;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = cb * COUT_TILE + j * 16 + m;
      const size_t pb = ((size_t)n * a.Cout + co) * (size_t)HW;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int px = p0 + 16 * i + 4 * k;
        if (EPI == 5) {
          if (u < U && co < a.Cout && px < HW) {
            const int y = px / a.Ws, x = px - y * a.Ws, Wo = 2 * a.Ws;           // (a quad never crosses a row: Ws % 4 == 0)
            const float sc = mk_sc[j], sh = mk_sh[j];
            const float v0 = acc[i][j][0] + bias_v[j], v1 = acc[i][j][1] + bias_v[j], v2 = acc[i][j][2] + bias_v[j], v3 = acc[i][j][3] + bias_v[j];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
              const size_t off = 4 * pb + (size_t)(2 * y + dy) * Wo + 2 * x;
              const float4 t0 = *reinterpret_cast<const float4*>(a.mk_u + off), t1 = *reinterpret_cast<const float4*>(a.mk_u + off + 4);
              float4 o0, o1;
              o0.x = leaky((sc * t0.x + sh) + v0, a.mk_slope); o0.y = leaky((sc * t0.y + sh) + v0, a.mk_slope);
              o0.z = leaky((sc * t0.z + sh) + v1, a.mk_slope); o0.w = leaky((sc * t0.w + sh) + v1, a.mk_slope);
              o1.x = leaky((sc * t1.x + sh) + v2, a.mk_slope); o1.y = leaky((sc * t1.y + sh) + v2, a.mk_slope);
              o1.z = leaky((sc * t1.z + sh) + v3, a.mk_slope); o1.w = leaky((sc * t1.w + sh) + v3, a.mk_slope);
              *reinterpret_cast<float4*>(a.out + off) = o0;
              *reinterpret_cast<float4*>(a.out + off + 4) = o1;
            }
          }
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          continue;
        }
        if (u < U && co < a.Cout && px < HW) {
          float4 o = make_float4(acc[i][j][0] + bias_v[j], acc[i][j][1] + bias_v[j], acc[i][j][2] + bias_v[j], acc[i][j][3] + bias_v[j]);
          if (EPI == 4) {
            const float4 t = *reinterpret_cast<const float4*>(a.mk_u + pb + px);
            const float sc = mk_sc[j], sh = mk_sh[j];
            o.x = leaky((sc * t.x + sh) + o.x, a.mk_slope); o.y = leaky((sc * t.y + sh) + o.y, a.mk_slope);
            o.z = leaky((sc * t.z + sh) + o.z, a.mk_slope); o.w = leaky((sc * t.w + sh) + o.w, a.mk_slope);
          }
          *reinterpret_cast<float4*>(a.out + pb + px) = o;
        }
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  int item = vb, chunk = 0, grp, cb, ring = 0;
  decode(item, grp, cb);
  load_cb(cb);
  lds_barrier();                                      // barrier #0
  // side jobs, while the staging waves fetch the first chunks
  if (a.ride_out != nullptr)
    for (int c = (int)blockIdx.x * 4 + wave; c < a.ride_C; c += 4 * (int)gridDim.x) conv_ride(a, c, lane);
  if (xf_epi) { xfin_header(a, xf_tag, xf_nparts); xfin_produce(a, xf_tag, xf_nparts); }
  lds_barrier();                                      // barrier #1: chunk 0 is in buffer 0
  for (int p = 0; p < T; ++p) {
    compute(smem + ring * BUF);
    if (++ring == 3) ring = 0;
    if (chunk + 1 == nchunks) {
      epilogue(grp, cb);
      chunk = 0; item += gridDim.x;
      if (p + 1 < T) { decode(item, grp, cb); load_cb(cb); }
    } else {
      ++chunk;
    }
    if (p + 1 < T) lds_barrier();
  }
}

int conv_k1g_switch();      // ms_conv.hip: option "conv.k1g" (0: off - A/B runs and the same-bits tests)
inline bool conv_k1g_eligible(const ConvArgs& a, int ks, int stride, int fetch) {
  if (conv_k1g_switch() == 0 || ks != 1 || stride != 1 || fetch != FETCH_NORMAL || a.act_bf16 != 0 || a.pro_mode != 0 || a.in2 != nullptr || a.stats != nullptr ||
      !(a.epi_mode == 0 || a.epi_mode == 4 || a.epi_mode == 5)) return false;
  if (a.xf_tab != nullptr && a.epi_mode == 0) return false;
  const long HW = (long)a.Hs * a.Ws;
  // measured scope: the channel-heavy levels (tools/ab_k1.py) - and, from 64 channels, rows that are not a multiple of 4 pixels (14 x 14: the deepest level of the
  // reference's shipped 224-pixel Prostate workload), where the tiled kernel falls to its scalar staging path (27-33 us per launch against 9-11 us at 16 x 16)
  const int cmin = (a.Ws % 4 != 0) ? 64 : 256;
  if (a.Cin % 8 != 0 || a.Cin < cmin || HW % 4 != 0 || (a.epi_mode == 5 && a.Ws % 4 != 0)) return false;
  if ((long long)a.N * a.Cin * HW * 4 >= (1LL << 31) || (long long)a.cin_pad * a.cout_pad * 4 >= (1LL << 31)) return false;
  return aligned16(a.in) && aligned16(a.out) && aligned16(a.w) && (a.epi_mode == 0 || aligned16(a.mk_u));
}
inline int conv_k1g_nt(const ConvArgs& a) {
  const long groups = cdiv((long)a.N * cdiv((long)a.Hs * a.Ws, 64L), 4L);
  for (int cand : {4, 2}) if (a.Cout >= 16 * cand && 2 * groups * cdiv(a.Cout, 16 * cand) >= 3L * num_cus()) return cand;
  return 1;
}

template <int NT, int EPI>
int launch_conv_k1g_t(ConvArgs a, hipStream_t st) {
  using G = K1gGeo<NT>;
  const size_t lds_bytes = sizeof(float) * 3 * (size_t)G::BUF;
  a.ncb = cdiv(a.Cout, 16 * NT);
  const long groups = cdiv((long)a.N * cdiv((long)a.Hs * a.Ws, 64L), 4L);
  const long nitems = groups * a.ncb;
  const int per_cu = std::min(conv_resident_per_cu((const void*)conv_k1g_kernel<NT, EPI>, lds_bytes), 4);
  long nblocks = std::min<long>(nitems, (long)num_cus() * per_cu);
  if (nblocks > a.ncb) nblocks -= nblocks % a.ncb;
  MS_LAUNCH((conv_k1g_kernel<NT, EPI>), dim3((unsigned)nblocks), dim3(512), lds_bytes, st, a);
  return check_launch("conv_k1g");
}

int conv_dispatch_k1g(const ConvArgs& a, hipStream_t st);      // ms_conv_inst_k1g.hip

}  // namespace ms
