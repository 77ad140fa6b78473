// Instantiations: second generation of the 3x3 stride-1 convolution on rows of 12 / 14 / 16 pixels (ms_conv_k3n.h), prologue-free and BatchNorm-apply variants.
#include "ms_conv_k3n.h"
namespace ms {
int conv_dispatch_k3n_p2(const ConvArgs& a, int mt, hipStream_t st);      // ms_conv_inst_k3n2.hip: the two-tensor prologue
template <int W>
static int k3n_w(const ConvArgs& a, int mt, hipStream_t st) {
  if (a.pro_mode == 0) return mt == 2 ? launch_conv_k3n_t<W, 2, 0>(a, st) : launch_conv_k3n_t<W, 1, 0>(a, st);
  return mt == 2 ? launch_conv_k3n_t<W, 2, 1>(a, st) : launch_conv_k3n_t<W, 1, 1>(a, st);
}
int conv_dispatch_k3n_k1(const ConvArgs& a, int mt, hipStream_t st);      // ms_conv_inst_k3n2.hip: the 1x1 variants (rows of 14 pixels)
int conv_dispatch_k3n_s2(const ConvArgs& a, int mt, hipStream_t st);      // ms_conv_inst_k3n2.hip: the stride-2 variants
int conv_dispatch_k3n(const ConvArgs& a, int ks, hipStream_t st, int stride) {
  const int mt = conv_k3n_mt(a);
  if (stride == 2) return conv_dispatch_k3n_s2(a, mt, st);
  if (ks == 1) return conv_dispatch_k3n_k1(a, mt, st);
  if (a.pro_mode == 2) return conv_dispatch_k3n_p2(a, mt, st);
  switch (a.Ws) {
    case 12: return k3n_w<12>(a, mt, st);
    case 14: return k3n_w<14>(a, mt, st);
    default: return k3n_w<16>(a, mt, st);
  }
}
}  // namespace ms

// Diagnostics (include/maxstyle_hip.h, MS_INTERNAL): L plain 128->128-style 3x3 layers on rows of 16 pixels as ONE persistent launch with grid barriers between the
// layers, ping-ponging between `a` and `b` (layer l reads (l even ? a : b), writes the other): the layer-chain go / no-go probe of tools/chain_probe.py.
// layers_dev: device scratch of ms_diag_k3n_chain_bytes(L) bytes; arrive: two zero-initialised device words that persist between calls; err: time-out word.
#include <vector>
#include "maxstyle_hip.h"
extern "C" size_t ms_diag_k3n_chain_bytes(int L) { return (size_t)(L < 1 ? 1 : L) * sizeof(ms::ConvArgs); }
extern "C" int ms_diag_k3n_chain(const float* a_buf, float* b_buf, const float* w_packed, int N, int C, int H, int L, void* layers_dev, unsigned* arrive, int* err, void* stream) {
  using namespace ms;
  if (N < 1 || C < 16 || C % 16 != 0 || H < 1 || L < 1 || L > 64 || (H * 16) % 128 != 0) { set_error("ms_diag_k3n_chain: C %% 16 == 0, rows of 16 pixels, H * 16 %% 128 == 0, 1 <= L <= 64"); return MS_ERR_INVALID; }
  static thread_local std::vector<ConvArgs> host;
  host.assign((size_t)L, ConvArgs{});
  for (int l = 0; l < L; ++l) {
    ConvArgs& a = host[(size_t)l];
    a.in = (l % 2 == 0) ? a_buf : b_buf; a.out = (l % 2 == 0) ? b_buf : const_cast<float*>(a_buf); a.w = w_packed;
    a.N = N; a.Cin = C; a.Hs = H; a.Ws = 16; a.Hin = H; a.Win = 16; a.Hout = H; a.Wout = 16; a.Cout = C; a.cout_real = C;
    a.cin_pad = C; a.cout_pad = (C + 63) / 64 * 64; a.ncb = C / 16; a.pro_cstride = 1; a.slope = 1.f;
  }
  hipStream_t st = (hipStream_t)stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap);          // (under capture the list of an earlier identical call is reused: a pageable copy cannot be captured)
  if (cap == hipStreamCaptureStatusNone && hipMemcpyAsync(layers_dev, host.data(), (size_t)L * sizeof(ConvArgs), hipMemcpyHostToDevice, st) != hipSuccess) { set_error("ms_diag_k3n_chain: copy of the layer list failed"); return MS_ERR_WORKSPACE; }
  using G = K3nGeo<16, 2>;
  const size_t lds_bytes = sizeof(float) * (3 * (size_t)G::BUF + 4 * (size_t)C);
  static std::once_flag attr_once;
  std::call_once(attr_once, []() { (void)hipFuncSetAttribute((const void*)conv_k3n_chain_kernel<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)); });
  const long nitems = (long)N * cdiv(H * 16, G::PIX) * (C / 16);
  const int per_cu = std::max(1, std::min(conv_resident_per_cu((const void*)conv_k3n_chain_kernel<16, 2>, lds_bytes), 2));
  long nblocks = std::min<long>(nitems, (long)num_cus() * per_cu);      // every workgroup resident: the grid barrier's premise
  if (nblocks > C / 16) nblocks -= nblocks % (C / 16);
  MS_LAUNCH((conv_k3n_chain_kernel<16, 2>), dim3((unsigned)nblocks), dim3(512), lds_bytes, st, (const ConvArgs*)layers_dev, L, arrive, err);
  return check_launch("conv_k3n_chain");
}
