// Weight gradients of the convolutions for the OUTER update: what autograd's convolution_backward computes for
// `weight` when loss.backward() runs after standard_training / hard_example_traininng (train_adv_supervised_segmentation_triplet.py:532-535;
// layers of encoder_decoder.py:22-74, 289-357, 423-482).  The inner loop never needs them (every network parameter is frozen there).
// Kernel: ms_wgrad_kernel.h.  This file: shape -> tile configuration, the fixed-order reduction of the partials, the C ABI.
#include <cstdlib>
#include "ms_wgrad_kernel.h"
#include "maxstyle_hip.h"

namespace ms {

struct WgPlan { int ab, bb, tw; bool vec; int nslots; };

static WgPlan wgrad_plan(int N, int M, int Nq, int Hp, int Wp, int stride, bool vec) {
  WgPlan p{};
  p.vec = vec;
  const bool big = (M > 16) || (Nq > 16);
  if (stride == 2 || !vec) p.tw = 16;
  else p.tw = (Wp >= 64) ? 64 : (Wp > 16 ? 32 : 16);
  // 64-wide tiles only for the 16x16-channel block (two tiles of every operand live in the producers' registers: wider channel blocks
  // at that width spill); 32-wide tiles take the exact 16/32 channel block shape, 16-wide ones 16x16 or 32x32
  if (p.tw == 64 && big) p.tw = 32;
  if (p.tw == 32) { p.ab = (M > 16) ? 2 : 1; p.bb = (Nq > 16) ? 2 : 1; }
  else p.ab = p.bb = big ? 2 : 1;
  const int ntiles = N * cdiv(Wp, p.tw) * cdiv(Hp, 4);
  const int npairs = cdiv(M, 16 * p.ab) * cdiv(Nq, 16 * p.bb);
  const int want = std::max(1, std::min(ntiles, 256 / npairs));
  const int per = cdiv(ntiles, want);
  p.nslots = cdiv(ntiles, per);
  return p;
}

#if !defined(MS_WGRAD_TU_A) && !defined(MS_WGRAD_TU_B) && !defined(MS_WGRAD_TU_C) && !defined(MS_WGRAD_TU_D)
// dW[i] (+)= sum over slots in a fixed order.  A workgroup owns 64 consecutive elements; its four waves each sum a quarter of the slots
// (8 independent loads in flight per lane - the chain of dependent L2 round trips was what made the naive loop 13 us), then combine in LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nslots, size_t numel, float* __restrict__ dw, int accumulate) {
  __shared__ float red[4][64];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + e;
  const int per = (nslots + 3) / 4;
  const int s0 = q * per, s1 = min(nslots, s0 + per);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (i < numel) {
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += partial[(size_t)(s + j) * numel + i];
    }
    for (; s < s1; ++s) acc[0] += partial[(size_t)s * numel + i];
  }
  red[q][e] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (q == 0 && i < numel) {
    const float t = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

// The same reduction for MANY weight tensors in one launch (a backward pass produces ~70 partial sets; one 5 us launch each was 10 % of the
// pass).  One 64-element block of one tensor per workgroup; the descriptor table lives on the device (addresses are stable across iterations).
struct WgBatchDesc { long long block_begin; const float* partial; float* dst; int numel, nslots, accumulate, pad; };   // 40 -> padded to 48 bytes
static_assert(sizeof(WgBatchDesc) == 40 || sizeof(WgBatchDesc) == 48, "descriptor layout");

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgBatchDesc* __restrict__ desc, int ndesc) {
  __shared__ float red[4][64];
  int lo = 0, hi = ndesc - 1;
  const long long blk = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].block_begin <= blk) lo = mid; else hi = mid - 1; }
  const WgBatchDesc d = desc[lo];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const size_t numel = (size_t)d.numel;
  const size_t i = (size_t)(blk - d.block_begin) * 64 + e;
  const int per = (d.nslots + 3) / 4;
  const int s0 = q * per, s1 = min(d.nslots, s0 + per);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (i < numel) {
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += d.partial[(size_t)(s + j) * numel + i];
    }
    for (; s < s1; ++s) acc[0] += d.partial[(size_t)s * numel + i];
  }
  red[q][e] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (q == 0 && i < numel) {
    const float t = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    d.dst[i] = d.accumulate ? d.dst[i] + t : t;
  }
}
#endif

// Prologue modes are template parameters (the staging code must not spend VALU cycles on identity math: it competes with the MFMA issue):
//   PM 2 / QM 1  conv3x3 whose input is BatchNorm+LeakyReLU of the previous conv (conv.3, inc.3, code_decoupler.3)
//   PM 2 / QM 0  conv followed by BatchNorm whose input is materialised (conv.0, inc.0, code_decoupler.0, final_conv.0)
//   PM 0 / QM 0  convs without BatchNorm (conv_input, down, up)
template <int KS, int S, bool VEC, bool QUPS, int PM, int QM>
static int dispatch_cfg(const WgArgs& a, const WgPlan& p, hipStream_t st) {
  const int key = p.tw * 100 + p.ab * 10 + p.bb;
  if constexpr (S == 2) {
    if (key == 1611) return launch_wgrad<KS, S, 1, 1, 16, VEC, false, PM, QM>(a, 256, st);
    if (key == 1622) return launch_wgrad<KS, S, 2, 2, 16, VEC, false, PM, QM>(a, 256, st);
  } else if constexpr (!VEC) {
    if (key == 1611) return launch_wgrad<KS, S, 1, 1, 16, false, QUPS, PM, QM>(a, 256, st);
    if (key == 1622) return launch_wgrad<KS, S, 2, 2, 16, false, QUPS, PM, QM>(a, 256, st);
  } else {
    switch (key) {
      case 6411: return launch_wgrad<KS, S, 1, 1, 64, true, QUPS, PM, QM>(a, 256, st);
      case 3221: return launch_wgrad<KS, S, 2, 1, 32, true, QUPS, PM, QM>(a, 256, st);
      case 3212: return launch_wgrad<KS, S, 1, 2, 32, true, QUPS, PM, QM>(a, 256, st);
      case 3211: return launch_wgrad<KS, S, 1, 1, 32, true, QUPS, PM, QM>(a, 256, st);
      case 3222: return launch_wgrad<KS, S, 2, 2, 32, true, QUPS, PM, QM>(a, 256, st);
      case 1611: return launch_wgrad<KS, S, 1, 1, 16, true, QUPS, PM, QM>(a, 256, st);
      case 1622: return launch_wgrad<KS, S, 2, 2, 16, true, QUPS, PM, QM>(a, 256, st);
    }
  }
  set_error("ms_conv_wgrad: no kernel for tile configuration %d", key);
  return MS_ERR_INVALID;
}

// implemented in the instantiation units; `a` arrives with p_mode in {0,2}, q_mode in {0,1}
int wgrad_dispatch_k3s1_act(const WgArgs& a, const WgPlan& p, bool qups, hipStream_t st);      // PM 2, QM 1
int wgrad_dispatch_k3s1_bn(const WgArgs& a, const WgPlan& p, bool qups, hipStream_t st);       // PM 2, QM 0
int wgrad_dispatch_k3s1_plain(const WgArgs& a, const WgPlan& p, bool qups, hipStream_t st);    // PM 0, QM 0
int wgrad_dispatch_k1s1(const WgArgs& a, const WgPlan& p, hipStream_t st);                     // PM 0|2, QM 0
int wgrad_dispatch_s2(const WgArgs& a, const WgPlan& p, int ks, hipStream_t st);               // PM 0, QM 0

#if defined(MS_WGRAD_TU_A)
int wgrad_dispatch_k3s1_act(const WgArgs& a, const WgPlan& p, bool qups, hipStream_t st) {
  if (p.vec) return qups ? dispatch_cfg<3, 1, true, true, 2, 1>(a, p, st) : dispatch_cfg<3, 1, true, false, 2, 1>(a, p, st);
  return qups ? dispatch_cfg<3, 1, false, true, 2, 1>(a, p, st) : dispatch_cfg<3, 1, false, false, 2, 1>(a, p, st);
}
#elif defined(MS_WGRAD_TU_C)
int wgrad_dispatch_k3s1_bn(const WgArgs& a, const WgPlan& p, bool qups, hipStream_t st) {
  if (p.vec) return qups ? dispatch_cfg<3, 1, true, true, 2, 0>(a, p, st) : dispatch_cfg<3, 1, true, false, 2, 0>(a, p, st);
  return qups ? dispatch_cfg<3, 1, false, true, 2, 0>(a, p, st) : dispatch_cfg<3, 1, false, false, 2, 0>(a, p, st);
}
#elif defined(MS_WGRAD_TU_D)
int wgrad_dispatch_k3s1_plain(const WgArgs& a, const WgPlan& p, bool qups, hipStream_t st) {
  if (p.vec) return qups ? dispatch_cfg<3, 1, true, true, 0, 0>(a, p, st) : dispatch_cfg<3, 1, true, false, 0, 0>(a, p, st);
  return qups ? dispatch_cfg<3, 1, false, true, 0, 0>(a, p, st) : dispatch_cfg<3, 1, false, false, 0, 0>(a, p, st);
}
#elif defined(MS_WGRAD_TU_B)
int wgrad_dispatch_k1s1(const WgArgs& a, const WgPlan& p, hipStream_t st) {
  if (a.p_mode == 2) return p.vec ? dispatch_cfg<1, 1, true, false, 2, 0>(a, p, st) : dispatch_cfg<1, 1, false, false, 2, 0>(a, p, st);
  return p.vec ? dispatch_cfg<1, 1, true, false, 0, 0>(a, p, st) : dispatch_cfg<1, 1, false, false, 0, 0>(a, p, st);
}
int wgrad_dispatch_s2(const WgArgs& a, const WgPlan& p, int ks, hipStream_t st) {
  if (ks == 3) return p.vec ? dispatch_cfg<3, 2, true, false, 0, 0>(a, p, st) : dispatch_cfg<3, 2, false, false, 0, 0>(a, p, st);
  return p.vec ? dispatch_cfg<2, 2, true, false, 0, 0>(a, p, st) : dispatch_cfg<2, 2, false, false, 0, 0>(a, p, st);
}
#endif
#if defined(MS_WGRAD_TU_A) || defined(MS_WGRAD_TU_B) || defined(MS_WGRAD_TU_C) || defined(MS_WGRAD_TU_D)
}  // namespace ms
#else

static bool wgrad_geometry_ok(int Hp, int Wp, int Hq, int Wq, int ks, int stride, int q_fetch) {
  const int HqL = q_fetch ? 2 * Hq : Hq, WqL = q_fetch ? 2 * Wq : Wq;
  if (ks == 3 && stride == 1) return Hp == HqL && Wp == WqL;
  if (ks == 1 && stride == 1) return Hp == HqL && Wp == WqL && !q_fetch;
  if (ks == 3 && stride == 2) return !q_fetch && Hp == (Hq + 1) / 2 && Wp == (Wq + 1) / 2;
  if (ks == 2 && stride == 2) return !q_fetch && Hq == 2 * Hp && Wq == 2 * Wp;
  return false;
}

static bool wgrad_vec(const float* p, const float* p2, const float* q, int Wp, int Wq, int q_fetch) {
  return (Wp % 4 == 0) && (q_fetch ? (Wq % 2 == 0) : (Wq % 4 == 0)) && aligned16(p) && (p2 == nullptr || aligned16(p2)) && aligned16(q);
}

}  // namespace ms
using namespace ms;

extern "C" size_t ms_conv_wgrad_ws_bytes(int N, int M, int Nq, int Hp, int Wp, int ks, int stride) {
  // the vector and the scalar staging paths pick different tiles: size for the larger slot count
  const WgPlan a = wgrad_plan(N, M, Nq, Hp, Wp, stride, true), b = wgrad_plan(N, M, Nq, Hp, Wp, stride, false);
  return (size_t)std::max(a.nslots, b.nslots) * M * Nq * ks * ks * sizeof(float);
}

static int wgrad_partials(const float* p, const float* p2, const float* q, int N, int M, int Nq, int Hp, int Wp, int Hq, int Wq,
                          int ks, int stride, int q_fetch, int p_mode, const float* pa, const float* pb, const float* pc,
                          int q_mode, const float* qa, const float* qb, int coef_stride, float slope,
                          void* ws, size_t ws_bytes, int* nslots_out, void* stream) {
  if (N < 1 || M < 1 || Nq < 1 || Hp < 1 || Wp < 1 || Hq < 1 || Wq < 1) { set_error("ms_conv_wgrad: invalid shape"); return MS_ERR_INVALID; }
  if (!wgrad_geometry_ok(Hp, Wp, Hq, Wq, ks, stride, q_fetch)) {
    set_error("ms_conv_wgrad: unsupported geometry (ks=%d stride=%d fetch=%d P %dx%d Q %dx%d)", ks, stride, q_fetch, Hp, Wp, Hq, Wq);
    return MS_ERR_INVALID;
  }
  if ((long long)M * Hp * Wp >= (1LL << 31) || (long long)Nq * Hq * Wq >= (1LL << 31)) { set_error("ms_conv_wgrad: one image exceeds 2^31 elements"); return MS_ERR_INVALID; }
  if ((p_mode != 0 && p_mode != 2) || (q_mode != 0 && q_mode != 1)) { set_error("ms_conv_wgrad: invalid prologue mode"); return MS_ERR_INVALID; }
  if (p_mode == 2 && (p2 == nullptr || pa == nullptr || pb == nullptr || pc == nullptr)) { set_error("ms_conv_wgrad: p_mode 2 needs p2 and three coefficient arrays"); return MS_ERR_INVALID; }
  if (q_mode == 1 && (qa == nullptr || qb == nullptr)) { set_error("ms_conv_wgrad: q_mode 1 needs two coefficient arrays"); return MS_ERR_INVALID; }
  const bool vec = wgrad_vec(p, p_mode == 2 ? p2 : nullptr, q, Wp, Wq, q_fetch);
  const WgPlan plan = wgrad_plan(N, M, Nq, Hp, Wp, stride, vec);
  const size_t numel = (size_t)M * Nq * ks * ks;
  if (ws == nullptr || ws_bytes < (size_t)plan.nslots * numel * sizeof(float)) { set_error("ms_conv_wgrad: workspace too small (see ms_conv_wgrad_ws_bytes)"); return MS_ERR_WORKSPACE; }
  WgArgs a{};
  a.p = p; a.p2 = p2; a.q = q; a.partial = (float*)ws;
  a.pa = pa; a.pb = pb; a.pc = pc; a.qa = qa; a.qb = qb;
  a.N = N; a.M = M; a.Nq = Nq; a.Hp = Hp; a.Wp = Wp; a.Hq = Hq; a.Wq = Wq;
  if (q_mode == 1 && !(slope >= 0.f && slope <= 1.f)) { set_error("ms_wgrad: activation slope outside [0, 1]"); return MS_ERR_INVALID; }
  a.p_mode = p_mode; a.q_mode = q_mode; a.coef_stride = coef_stride < 1 ? 1 : coef_stride; a.slope = slope;
  a.dbg = opt(OPT_DIAG_CONV_DBG);
  a.trace = wgrad_trace_buffer();      // cycle-stamp builds only (ms_diag_set_trace)
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (ks == 3 && stride == 1) {
    if (p_mode == 0 && q_mode == 1) { set_error("ms_conv_wgrad: q_mode 1 is built together with p_mode 2 only"); return MS_ERR_INVALID; }
    if (p_mode == 0) rc = wgrad_dispatch_k3s1_plain(a, plan, q_fetch != 0, st);
    else rc = (q_mode == 1) ? wgrad_dispatch_k3s1_act(a, plan, q_fetch != 0, st) : wgrad_dispatch_k3s1_bn(a, plan, q_fetch != 0, st);
  } else if (ks == 1) {
    if (q_mode != 0) { set_error("ms_conv_wgrad: the 1x1 kernels are built for q_mode 0"); return MS_ERR_INVALID; }
    rc = wgrad_dispatch_k1s1(a, plan, st);
  } else {
    if (p_mode != 0 || q_mode != 0) { set_error("ms_conv_wgrad: the stride-2 kernels are built without prologues"); return MS_ERR_INVALID; }
    rc = wgrad_dispatch_s2(a, plan, ks, st);
  }
  if (rc != MS_OK) return rc;
  *nslots_out = plan.nslots;
  return MS_OK;
}

extern "C" int ms_conv_wgrad(const float* p, const float* p2, const float* q, float* dw, int N, int M, int Nq, int Hp, int Wp, int Hq, int Wq,
                             int ks, int stride, int q_fetch, int p_mode, const float* pa, const float* pb, const float* pc,
                             int q_mode, const float* qa, const float* qb, int coef_stride, float slope, int accumulate,
                             void* ws, size_t ws_bytes, void* stream) {
  int nslots = 0;
  if (int rc = wgrad_partials(p, p2, q, N, M, Nq, Hp, Wp, Hq, Wq, ks, stride, q_fetch, p_mode, pa, pb, pc, q_mode, qa, qb, coef_stride, slope, ws, ws_bytes, &nslots, stream)) return rc;
  const size_t numel = (size_t)M * Nq * ks * ks;
  MS_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((numel + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nslots, numel, dw, accumulate);
  return check_launch("wgrad_reduce");
}

extern "C" int ms_conv_wgrad_partials(const float* p, const float* p2, const float* q, int N, int M, int Nq, int Hp, int Wp, int Hq, int Wq,
                                      int ks, int stride, int q_fetch, int p_mode, const float* pa, const float* pb, const float* pc,
                                      int q_mode, const float* qa, const float* qb, int coef_stride, float slope,
                                      void* ws, size_t ws_bytes, int* nslots_out, void* stream) {
  if (nslots_out == nullptr) { set_error("ms_conv_wgrad_partials: nslots_out is NULL"); return MS_ERR_INVALID; }
  return wgrad_partials(p, p2, q, N, M, Nq, Hp, Wp, Hq, Wq, ks, stride, q_fetch, p_mode, pa, pb, pc, q_mode, qa, qb, coef_stride, slope, ws, ws_bytes, nslots_out, stream);
}

extern "C" size_t ms_wgrad_batch_desc_bytes(void) { return sizeof(WgBatchDesc); }

extern "C" int ms_wgrad_reduce_batch(const void* desc_dev, int ndesc, long long total_blocks, void* stream) {
  if (ndesc < 1 || total_blocks < 1 || desc_dev == nullptr) { set_error("ms_wgrad_reduce_batch: nothing to do"); return MS_ERR_INVALID; }
  MS_LAUNCH(wgrad_reduce_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const WgBatchDesc*)desc_dev, ndesc);
  return check_launch("wgrad_reduce_batch");
}
#endif
