// Weight gradients of the convolutions for the OUTER update: what autograd's convolution_backward computes for
// `weight` when loss.backward() runs after standard_training / hard_example_traininng (train_adv_supervised_segmentation_triplet.py:532-535;
// layers of encoder_decoder.py:22-74, 289-357, 423-482).  The inner loop never needs them (every network parameter is frozen there).
// Kernel: ms_wgrad_kernel.h.  This file: shape -> tile configuration, the fixed-order reduction of the partials, the C ABI.
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
  if (p.tw == 64) {
    p.ab = (M > 16) ? 2 : 1; p.bb = (Nq > 16) ? 2 : 1;
    if (p.ab == 2 && p.bb == 2) p.tw = 32;
  } else {
    p.ab = p.bb = big ? 2 : 1;
  }
  const int ntiles = N * cdiv(Wp, p.tw) * cdiv(Hp, 4);
  const int npairs = cdiv(M, 16 * p.ab) * cdiv(Nq, 16 * p.bb);
  const int want = std::max(1, std::min(ntiles, 256 / npairs));
  const int per = cdiv(ntiles, want);
  p.nslots = cdiv(ntiles, per);
  return p;
}

#if !defined(MS_WGRAD_TU_A) && !defined(MS_WGRAD_TU_B)
// dW[i] (+)= sum over slots, fixed order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nslots, size_t numel, float* __restrict__ dw, int accumulate) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= numel) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int s = 0;
  for (; s + 4 <= nslots; s += 4) {
    s0 += partial[(size_t)s * numel + i]; s1 += partial[(size_t)(s + 1) * numel + i];
    s2 += partial[(size_t)(s + 2) * numel + i]; s3 += partial[(size_t)(s + 3) * numel + i];
  }
  for (; s < nslots; ++s) s0 += partial[(size_t)s * numel + i];
  const float t = (s0 + s1) + (s2 + s3);
  dw[i] = accumulate ? dw[i] + t : t;
}
#endif

template <int KS, int S, bool VEC, bool QUPS>
static int dispatch_cfg(const WgArgs& a, const WgPlan& p, hipStream_t st) {
  const int key = p.tw * 100 + p.ab * 10 + p.bb;
  if constexpr (S == 2) {
    if (key == 1611) return launch_wgrad<KS, S, 1, 1, 16, VEC, false>(a, 256, st);
    if (key == 1622) return launch_wgrad<KS, S, 2, 2, 16, VEC, false>(a, 256, st);
  } else if constexpr (!VEC) {
    if (key == 1611) return launch_wgrad<KS, S, 1, 1, 16, false, QUPS>(a, 256, st);
    if (key == 1622) return launch_wgrad<KS, S, 2, 2, 16, false, QUPS>(a, 256, st);
  } else {
    switch (key) {
      case 6411: return launch_wgrad<KS, S, 1, 1, 64, true, QUPS>(a, 256, st);
      case 6421: return launch_wgrad<KS, S, 2, 1, 64, true, QUPS>(a, 256, st);
      case 6412: return launch_wgrad<KS, S, 1, 2, 64, true, QUPS>(a, 256, st);
      case 3211: return launch_wgrad<KS, S, 1, 1, 32, true, QUPS>(a, 256, st);
      case 3222: return launch_wgrad<KS, S, 2, 2, 32, true, QUPS>(a, 256, st);
      case 1611: return launch_wgrad<KS, S, 1, 1, 16, true, QUPS>(a, 256, st);
      case 1622: return launch_wgrad<KS, S, 2, 2, 16, true, QUPS>(a, 256, st);
    }
  }
  set_error("ms_conv_wgrad: no kernel for tile configuration %d", key);
  return MS_ERR_INVALID;
}

int wgrad_dispatch_k3s1(const WgArgs& a, const WgPlan& p, bool qups, hipStream_t st);
int wgrad_dispatch_k1s1(const WgArgs& a, const WgPlan& p, hipStream_t st);
int wgrad_dispatch_s2(const WgArgs& a, const WgPlan& p, int ks, hipStream_t st);

#if defined(MS_WGRAD_TU_A)
int wgrad_dispatch_k3s1(const WgArgs& a, const WgPlan& p, bool qups, hipStream_t st) {
  if (p.vec) return qups ? dispatch_cfg<3, 1, true, true>(a, p, st) : dispatch_cfg<3, 1, true, false>(a, p, st);
  return qups ? dispatch_cfg<3, 1, false, true>(a, p, st) : dispatch_cfg<3, 1, false, false>(a, p, st);
}
#elif defined(MS_WGRAD_TU_B)
int wgrad_dispatch_k1s1(const WgArgs& a, const WgPlan& p, hipStream_t st) {
  return p.vec ? dispatch_cfg<1, 1, true, false>(a, p, st) : dispatch_cfg<1, 1, false, false>(a, p, st);
}
int wgrad_dispatch_s2(const WgArgs& a, const WgPlan& p, int ks, hipStream_t st) {
  if (ks == 3) return p.vec ? dispatch_cfg<3, 2, true, false>(a, p, st) : dispatch_cfg<3, 2, false, false>(a, p, st);
  return p.vec ? dispatch_cfg<2, 2, true, false>(a, p, st) : dispatch_cfg<2, 2, false, false>(a, p, st);
}
#endif
#if defined(MS_WGRAD_TU_A) || defined(MS_WGRAD_TU_B)
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

extern "C" int ms_conv_wgrad(const float* p, const float* p2, const float* q, float* dw, int N, int M, int Nq, int Hp, int Wp, int Hq, int Wq,
                             int ks, int stride, int q_fetch, int p_mode, const float* pa, const float* pb, const float* pc,
                             int q_mode, const float* qa, const float* qb, int coef_stride, float slope, int accumulate,
                             void* ws, size_t ws_bytes, void* stream) {
  if (N < 1 || M < 1 || Nq < 1 || Hp < 1 || Wp < 1 || Hq < 1 || Wq < 1) { set_error("ms_conv_wgrad: invalid shape"); return MS_ERR_INVALID; }
  if (!wgrad_geometry_ok(Hp, Wp, Hq, Wq, ks, stride, q_fetch)) {
    set_error("ms_conv_wgrad: unsupported geometry (ks=%d stride=%d fetch=%d P %dx%d Q %dx%d)", ks, stride, q_fetch, Hp, Wp, Hq, Wq);
    return MS_ERR_INVALID;
  }
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
  a.p_mode = p_mode; a.q_mode = q_mode; a.coef_stride = coef_stride < 1 ? 1 : coef_stride; a.slope = slope;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (ks == 3 && stride == 1) rc = wgrad_dispatch_k3s1(a, plan, q_fetch != 0, st);
  else if (ks == 1) rc = wgrad_dispatch_k1s1(a, plan, st);
  else rc = wgrad_dispatch_s2(a, plan, ks, st);
  if (rc != MS_OK) return rc;
  MS_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, st, (const float*)ws, plan.nslots, numel, dw, accumulate);
  return check_launch("wgrad_reduce");
}
#endif
