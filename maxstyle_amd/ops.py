"""Thin tensor-level wrappers over the C ABI (pointer/size marshaling only; PyTorch is used for device
memory and the current HIP stream, nothing else)."""
import torch

from ._lib import lib, check

_ws_cache = {}


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw handle of torch's current stream on the current device (without building a torch.cuda.Stream object per launch)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    if t is None:
        return 0
    return t if isinstance(t, int) else t.data_ptr()


def coef_ptrs(coef4):
    """Raw pointers to columns 0,1,2 of an interleaved [C,4] coefficient record (use with pro_cstride=4)."""
    p = coef4.data_ptr()
    return p, p + 4, p + 8


def _need_cuda_f32(*ts):
    for t in ts:
        if t is None or isinstance(t, int):
            continue
        if not t.is_cuda:
            raise RuntimeError("maxstyle_amd ops run on the MI355X only: got a CPU tensor (there is no CPU fallback)")
        if t.dtype != torch.float32:
            raise TypeError(f"maxstyle_amd ops are fp32; got {t.dtype}")
        if not t.is_contiguous():
            raise ValueError("maxstyle_amd ops need contiguous NCHW tensors")


def _need_cuda_act(t):
    """An activation tensor: fp32, or bf16 storage (the `*_bf16` entry points: statistics and arithmetic stay fp32)."""
    if not t.is_cuda:
        raise RuntimeError("maxstyle_amd ops run on the MI355X only: got a CPU tensor (there is no CPU fallback)")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"maxstyle_amd activations are fp32 or bf16; got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("maxstyle_amd ops need contiguous NCHW tensors")
    return t.dtype == torch.bfloat16


def _acts(*ts):
    """Activation tensors of one call: all fp32 or all bf16 (the conv stack's `_bf16` entry points). Returns True for bf16."""
    flags = {_need_cuda_act(t) for t in ts if t is not None and not isinstance(t, int)}
    if len(flags) > 1:
        raise TypeError("activation tensors of one call must share a storage type (all fp32 or all bf16)")
    return bool(flags and flags.pop())


def _fn(name, bf16, mfma_bf16=False):
    if bf16 and mfma_bf16:
        return getattr(lib, name + "_bf16m")
    return getattr(lib, name + "_bf16") if bf16 else getattr(lib, name)


def workspace(nbytes, device):
    """A cached scratch buffer per (device, stream); grown geometrically. Kernels on one stream are ordered, so reuse is safe."""
    key = (device, _stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_style_ws_cache = {}


def style_ws(B, C, HW, device, kind="ws"):
    """Workspace of one MaxStyle layer shape: zero-filled once, dedicated to that shape on that stream (its tail is the persistent
    launch-epoch state of the single-read kernel: include/maxstyle_hip.h, ms_style_ws_bytes).  kind="fused": the state block alone."""
    key = (device, _stream(), B, C, HW, kind)
    buf = _style_ws_cache.get(key)
    if buf is None:
        n = {"ws": lib.ms_style_ws_bytes, "fused": lib.ms_style_fused_ws_bytes, "ws_bf16": lib.ms_style_ws_bytes_bf16,
             "fused_bf16": lib.ms_style_fused_ws_bytes_bf16}[kind](B, C, HW)
        buf = torch.zeros(max(int(n), 64), dtype=torch.uint8, device=device)
        _style_ws_cache[key] = buf
    return buf


def style_moments(x, eps=1e-6):
    _need_cuda_f32(x)
    B, C = x.shape[:2]
    HW = x[0, 0].numel()
    mu = torch.empty(B, C, 1, 1, device=x.device, dtype=torch.float32)
    sig = torch.empty_like(mu)
    ws = style_ws(B, C, HW, x.device)
    check(lib.ms_style_moments(x.data_ptr(), mu.data_ptr(), sig.data_ptr(), B * C, HW, eps, ws.data_ptr(), ws.numel(), _stream()), "ms_style_moments")
    return mu, sig


def style_fwd(x, perm, lmda, gamma_noise, beta_noise, gamma_std, beta_std, compute_std, eps=1e-6, out=None, impl=None):
    """Returns y, mu, sig, coefA, coefS. gamma_std/beta_std are [1,C,1,1] buffers (written if compute_std).
    impl: None = library dispatch, "fused" = single-read kernel, "3k" = three-launch path.  x may be bf16 (activation storage): y is bf16 then,
    everything else fp32."""
    bf16 = _need_cuda_act(x)
    _need_cuda_f32(lmda, gamma_noise, beta_noise, gamma_std, beta_std)      # compute_std: bool, or the flag word (bit 0 std, bit 1 no clamp)
    B, C = x.shape[:2]
    HW = x[0, 0].numel()
    dev = x.device
    y = torch.empty_like(x) if out is None else out
    if y.dtype != x.dtype:
        raise TypeError("style_fwd: out must have the dtype of x")
    stats = torch.empty(4, B, C, 1, 1, device=dev, dtype=torch.float32)
    mu, sig, cA, cS = stats[0], stats[1], stats[2], stats[3]
    if bf16:
        if impl == "3k":
            raise NotImplementedError("bf16 storage: library dispatch or the single-read kernel")
        ws = style_ws(B, C, HW, dev, "fused_bf16" if impl == "fused" else "ws_bf16")
        fn = lib.ms_style_fwd_fused_bf16 if impl == "fused" else lib.ms_style_fwd_bf16
    else:
        ws = style_ws(B, C, HW, dev, "fused" if impl == "fused" else "ws")
        fn = {None: lib.ms_style_fwd, "fused": lib.ms_style_fwd_fused, "3k": lib.ms_style_fwd_3k}[impl]
    check(fn(x.data_ptr(), y.data_ptr(), mu.data_ptr(), sig.data_ptr(), gamma_std.data_ptr(), beta_std.data_ptr(),
             int(compute_std), _ptr(lmda), _ptr(gamma_noise), _ptr(beta_noise), _ptr(perm),
             cA.data_ptr(), cS.data_ptr(), B, C, HW, eps, ws.data_ptr(), ws.numel(), _stream()), "ms_style_fwd")
    return y, mu, sig, cA, cS


def style_bwd(dy, x, mu, sig, coefA, gamma_std, beta_std, lmda, perm, need_dx, need_noise, need_lmda):
    bf16 = _need_cuda_act(x)
    if _need_cuda_act(dy) != bf16:
        raise TypeError("style_bwd: dy and x must share one storage type")
    B, C = x.shape[:2]
    HW = x[0, 0].numel()
    dev = x.device
    dx = torch.empty_like(x) if need_dx else None
    dg = torch.empty(B, C, 1, 1, device=dev, dtype=torch.float32) if need_noise else None
    db = torch.empty(B, C, 1, 1, device=dev, dtype=torch.float32) if need_noise else None
    dl = torch.empty(B, 1, 1, 1, device=dev, dtype=torch.float32) if need_lmda else None
    ws = style_ws(B, C, HW, dev, "ws_bf16" if bf16 else "ws")
    fn = lib.ms_style_bwd_bf16 if bf16 else lib.ms_style_bwd
    check(fn(dy.data_ptr(), x.data_ptr(), _ptr(dx), mu.data_ptr(), sig.data_ptr(), coefA.data_ptr(),
             _ptr(gamma_std), _ptr(beta_std), _ptr(lmda), _ptr(perm), _ptr(dg), _ptr(db), _ptr(dl),
             B, C, HW, ws.data_ptr(), ws.numel(), _stream()), "ms_style_bwd")
    return dx, dg, db, dl


def adam_step(p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-8, step_dev=None):
    _need_cuda_f32(p, g, m, v)
    check(lib.ms_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, b1, b2, eps, int(step),
                           _ptr(step_dev), _stream()), "ms_adam_step")


# ---------------------------------------------------------------------------------------------------------
# convolution stack
# ---------------------------------------------------------------------------------------------------------
FETCH_NORMAL, FETCH_UPS2, FETCH_ZINS2 = 0, 1, 2
EPI_POOL2 = 6                # ms_conv2d epi_mode: 2x2-pooled store (MS_EPI_POOL2)
FETCH_WINO_BLOCKS = 0x1000  # MS_FETCH_WINO_BLOCKS: the block form of the Winograd kernel wherever legal (test / A-B switch)
FETCH_WINO_U = 0x800        # MS_FETCH_WINO_U: the packed weights carry the Winograd appendix (with_wino_appendix below)
FETCH_WINO_NT1 = 0x400      # MS_FETCH_WINO_NT1: with FETCH_WINOGRAD, the one-channel-block variant (A/B and test switch; same bits per output element)
FETCH_WINOGRAD = 0x100      # MS_FETCH_WINOGRAD: OR into fetch = the caller accepts the Winograd form of a 3x3 stride-1 convolution (include/maxstyle_hip.h)


def _pack(w4):
    """[Cout,Cin,k,k] -> [k*k][roundup(Cin,4)][roundup(Cout,64)] (zero padded), the layout ms_conv2d reads."""
    Cout, Cin, k, _ = w4.shape
    cin_pad = (Cin + 3) // 4 * 4
    cout_pad = (Cout + 63) // 64 * 64
    wp = torch.zeros(k * k, cin_pad, cout_pad, device=w4.device, dtype=torch.float32)
    wp[:, :Cin, :Cout] = w4.detach().float().permute(2, 3, 1, 0).reshape(k * k, Cin, Cout)
    return wp.contiguous()


def pack_conv_weight(w):
    """nn.Conv2d weight [Cout,Cin,k,k] for the forward convolution."""
    return _pack(w)


def pack_conv_weight_dgrad(w):
    """Data-gradient of nn.Conv2d(k,p=k//2): a convolution of dY with the flipped, in/out-swapped kernel
    (stride 1: as is; stride 2: on the zero-inserted dY, fetch=FETCH_ZINS2)."""
    return _pack(w.detach().flip(2, 3).transpose(0, 1).contiguous())


def pack_convT_weight(w):
    """nn.ConvTranspose2d(k=2,s=2) weight [Cin,Cout,2,2] as a k=1 GEMM with 4*Cout columns: col=(dy*2+dx)*Cout+co."""
    Cin, Cout = w.shape[:2]
    g = w.detach().float().permute(2, 3, 1, 0).reshape(4 * Cout, Cin)     # [(dy,dx,co), ci]
    return _pack(g.reshape(4 * Cout, Cin, 1, 1))


def pack_convT_weight_dgrad(w):
    """Data-gradient of ConvTranspose2d(k=2,s=2) = Conv2d(k=2,s=2,p=0) from Cout to Cin channels with kernel w[ci][co][dy][dx]."""
    return _pack(w.detach())          # as a conv weight: [Cout'=Cin, Cin'=Cout, 2, 2]


def with_wino_appendix(wp, Cin, Cout):
    """The packed 3x3 weights `wp` [9, cin_pad, cout_pad] re-homed into ONE buffer that also holds their Winograd appendix (include/maxstyle_hip.h, MS_FETCH_WINO_U):
    returns (view of the taps - same shape, the buffer's first bytes -, has_appendix).  Call wino_repack(view, Cin, Cout) after every change of the taps."""
    n_u = int(lib.ms_wino_pack_floats(Cin, Cout))
    if wp.dim() != 3 or wp.shape[0] != 9 or n_u == 0:
        return wp, False
    buf = torch.zeros(wp.numel() + n_u, dtype=torch.float32, device=wp.device)
    view = buf[:wp.numel()].view(wp.shape)
    view.copy_(wp)
    view._ms_wino_buf = buf                      # keeps the allocation alive with the view
    wino_repack(view, Cin, Cout)
    return view, True


def wino_repack(wp_view, Cin, Cout):
    check(lib.ms_wino_pack(wp_view.data_ptr(), Cin, Cout, _stream()), "ms_wino_pack")


def conv_out_hw(Hs, Ws, ks, stride, fetch):
    Hin, Win = (Hs, Ws) if fetch == FETCH_NORMAL else (2 * Hs, 2 * Ws)
    pad = 1 if ks == 3 else 0
    return (Hin + 2 * pad - ks) // stride + 1, (Win + 2 * pad - ks) // stride + 1


def conv2d(x, wp, bias, Cout, ks, stride=1, fetch=FETCH_NORMAL, pro_mode=0, pro_a=None, pro_b=None, pro_c=None, pro_nstride=0,
           pro_cstride=1, slope=1.0, epi_mode=0, out=None, stats=None, in2=None, mfma_bf16=False):
    """ms_conv2d wrapper (bf16 tensors: ms_conv2d_bf16, or ms_conv2d_bf16m with mfma_bf16=True). Returns out ([N,Cout,Hout,Wout], or [N,Cout,2H,2W] for the ConvTranspose epilogue)."""
    _need_cuda_f32(wp, bias, pro_a, pro_b, pro_c, stats)
    bf = _acts(x, in2, out)
    N, Cin, Hs, Ws = x.shape
    Ho, Wo = conv_out_hw(Hs, Ws, ks, stride, fetch & 0xFF)
    if out is None:
        shape = (N, Cout, 2 * Ho, 2 * Wo) if epi_mode == 2 else ((N, Cout, Ho // 2, Wo // 2) if epi_mode == EPI_POOL2 else (N, Cout, Ho, Wo))
        if epi_mode == 1:
            raise ValueError("accumulate epilogue needs an existing `out`")
        out = torch.empty(shape, device=x.device, dtype=x.dtype)
    check(_fn("ms_conv2d", bf, mfma_bf16)(x.data_ptr(), _ptr(in2), out.data_ptr(), wp.data_ptr(), _ptr(bias), N, Cin, Hs, Ws, Cout, ks, stride, fetch,
                        pro_mode, _ptr(pro_a), _ptr(pro_b), _ptr(pro_c), pro_nstride, pro_cstride, slope, epi_mode, _ptr(stats), _stream()), "ms_conv2d")
    return out


def conv2d_actbwd(x, wp, Cout, ks, u, coef4, act_slope, pro_mode=0, pro_a=None, pro_b=None, pro_c=None, pro_cstride=1, slope=1.0, in2=None, stride=1, mfma_bf16=False,
                  fetch=FETCH_NORMAL):
    """ms_conv2d_actbwd wrapper: conv (no bias) -> * LeakyReLU'(coef4.scale*u + coef4.shift) -> (out, tab); tab feeds bn_bwd_coefs(tab, 0, ...)."""
    _need_cuda_f32(wp, coef4, pro_a, pro_b, pro_c)
    bf = _acts(x, u, in2)
    N, Cin, Hs, Ws = x.shape
    Ho, Wo = conv_out_hw(Hs, Ws, ks, stride, FETCH_NORMAL)
    out = torch.empty((N, Cout, Ho, Wo), device=x.device, dtype=x.dtype)
    tab = torch.full((lib.ms_conv_actbwd_tab_bytes(Cout) // 4,), float("nan"), device=x.device, dtype=torch.float32)
    check(_fn("ms_conv2d_actbwd", bf, mfma_bf16)(x.data_ptr(), _ptr(in2), out.data_ptr(), wp.data_ptr(), N, Cin, Hs, Ws, Cout, ks, stride, fetch,
                               pro_mode, _ptr(pro_a), _ptr(pro_b), _ptr(pro_c), 0, pro_cstride, slope, u.data_ptr(), coef4.data_ptr(), act_slope,
                               tab.data_ptr(), _stream()), "ms_conv2d_actbwd")
    return out, tab


def conv_stats_buffer(N, Cout, Ho, Wo, device):
    parts = lib.ms_conv_stats_parts(N, Ho, Wo)
    return torch.empty(Cout * parts + 1, 4, device=device, dtype=torch.float32), parts


def bn_finalize(stats, nparts, gamma, beta, eps=1e-5, out=None):
    C = gamma.numel()
    coef = torch.empty(C, 4, device=gamma.device, dtype=torch.float32) if out is None else out
    check(lib.ms_bn_finalize(stats.data_ptr(), nparts, gamma.data_ptr(), beta.data_ptr(), eps, coef.data_ptr(), C, _stream()), "ms_bn_finalize")
    return coef


def bn_act(u, coef4, res=None, res_mode=0, slope=0.2, out=None):
    _need_cuda_f32(coef4)
    bf = _acts(u, res, out)
    N, C, H, W = u.shape
    out = torch.empty_like(u) if out is None else out
    check(_fn("ms_bn_act", bf)(u.data_ptr(), coef4.data_ptr(), _ptr(res), res_mode, out.data_ptr(), N, C, H, W, slope, _stream()), "ms_bn_act")
    return out


def act_bwd_reduce(gin, ref, u, coef4, slope, gout=None, part=None):
    _need_cuda_f32(coef4, part)
    bf = _acts(gin, ref, u, gout)
    N, C, H, W = u.shape
    nparts = lib.ms_act_bwd_parts(N, C, H * W)
    gout = torch.empty_like(gin) if gout is None else gout
    part = torch.empty(C, nparts, 2, device=u.device, dtype=torch.float32) if part is None else part
    check(_fn("ms_act_bwd_reduce", bf)(gin.data_ptr(), _ptr(ref), u.data_ptr(), coef4.data_ptr(), gout.data_ptr(), part.data_ptr(), N, C, H * W, slope, _stream()),
          "ms_act_bwd_reduce")
    return gout, part, nparts


def bn_bwd_coefs(part, nparts, coef4, count, out=None):
    C = coef4.shape[0]
    out = torch.empty(C, 4, device=coef4.device, dtype=torch.float32) if out is None else out
    check(lib.ms_bn_bwd_coefs(part.data_ptr(), nparts, coef4.data_ptr(), float(count), out.data_ptr(), C, _stream()), "ms_bn_bwd_coefs")
    return out


def pool2_sum(x, out=None, accumulate=False):
    bf = _acts(x, out)
    N, C, H, W = x.shape
    if out is None:
        out = torch.empty(N, C, H // 2, W // 2, device=x.device, dtype=x.dtype)
    check(_fn("ms_pool2_sum", bf)(x.data_ptr(), out.data_ptr(), N * C, H // 2, W // 2, 1 if accumulate else 0, _stream()), "ms_pool2_sum")
    return out


def head_fwd(h, w, b, apply_sigmoid, out=None):
    _need_cuda_f32(w, b)
    bf = _acts(h, out)
    N, C, H, W = h.shape
    K = w.shape[0]
    out = torch.empty(N, K, H, W, device=h.device, dtype=h.dtype) if out is None else out
    check(_fn("ms_head_fwd", bf)(h.data_ptr(), w.data_ptr(), _ptr(b), out.data_ptr(), N, C, K, H * W, 1 if apply_sigmoid else 0, _stream()), "ms_head_fwd")
    return out


def head_bwd(dout, out, w, C, apply_sigmoid, dh=None):
    _need_cuda_f32(w)
    bf = _acts(dout, out, dh)
    N, K, H, W = dout.shape
    dh = torch.empty(N, C, H, W, device=dout.device, dtype=dout.dtype) if dh is None else dh
    check(_fn("ms_head_bwd", bf)(dout.data_ptr(), _ptr(out), w.data_ptr(), dh.data_ptr(), N, C, K, H * W, 1 if apply_sigmoid else 0, _stream()), "ms_head_bwd")
    return dh


def head_ce(h, w, b, labels, loss_sign=1.0, need_dh=True, need_logits=False, dh=None, logits=None, loss_out=None, loss_slot_dev=None):
    """loss = loss_sign * cross_entropy_2D(w h + b, labels); returns (loss[1] device tensor, dh, logits)."""
    _need_cuda_f32(w, b, logits, loss_out)
    bf = _acts(h, dh)
    if labels.dtype != torch.int64 or not labels.is_cuda or not labels.is_contiguous():
        raise TypeError("labels must be a contiguous CUDA int64 tensor [N,H,W]")
    N, C, H, W = h.shape
    K = w.shape[0]
    dev = h.device
    if need_dh and dh is None:
        dh = torch.empty_like(h)
    if need_logits and logits is None:
        logits = torch.empty(N, K, H, W, device=dev, dtype=torch.float32)
    if loss_out is None:
        loss_out = torch.empty(1, device=dev, dtype=torch.float32)
    nbytes = lib.ms_head_ce_ws_bytes(N, H * W)
    ws = workspace(nbytes, dev)
    check(_fn("ms_head_ce", bf)(h.data_ptr(), w.data_ptr(), _ptr(b), labels.data_ptr(), _ptr(dh if need_dh else None), _ptr(logits if need_logits else None),
                         loss_out.data_ptr(), _ptr(loss_slot_dev), N, C, K, H * W, loss_sign, ws.data_ptr(), ws.numel(), _stream()), "ms_head_ce")
    return loss_out, dh, logits


def rescale_intensity(data, new_min=0.0, new_max=1.0, eps=1e-20):
    """common_utils/basic_operations.py:257-281 on the GPU (per (n,c) plane min-max)."""
    _need_cuda_f32(data)
    if data.dim() == 3:
        planes, hw = data.shape[0], data.shape[1] * data.shape[2]
    elif data.dim() >= 4:
        planes, hw = data.shape[0] * data.shape[1], data[0, 0].numel()
    else:
        raise ValueError
    out = torch.empty_like(data)
    check(lib.ms_rescale_intensity(data.data_ptr(), out.data_ptr(), planes, hw, new_min, new_max, eps, _stream()), "ms_rescale_intensity")
    return out


_wgrad_ws = {}


def conv_wgrad(p, q, ks, stride=1, q_fetch=0, p_bnbwd=None, q_act=None, out=None, accumulate=False):
    """weight.grad of a convolution (ms_conv_wgrad). p: gradient w.r.t. the conv output [N,M,Hp,Wp] (ConvTranspose2d: its input),
    q: conv input [N,Nq,Hq,Wq] (ConvTranspose2d: gradient w.r.t. its output).  p_bnbwd=(bcoef4 [M,4], u): P = a*p + b*u + c;
    q_act=(coef4 [Nq,4], slope): Q = LeakyReLU(a*q + b).  Returns dw [M, Nq, ks, ks]."""
    _need_cuda_f32(p); _need_cuda_f32(q)
    N, M, Hp, Wp = p.shape
    Nq, Hq, Wq = q.shape[1:]
    if out is None:
        out = torch.empty(M, Nq, ks, ks, dtype=torch.float32, device=p.device)
    nbytes = lib.ms_conv_wgrad_ws_bytes(N, M, Nq, Hp, Wp, ks, stride)
    key = (p.device, torch.cuda.current_stream().cuda_stream)
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=p.device)
        _wgrad_ws[key] = ws
    pm, p2, pa, pb, pc = 0, 0, 0, 0, 0
    qm, qa, qb, slope = 0, 0, 0, 1.0
    if p_bnbwd is not None:
        pm = 2
        pa, pb, pc = coef_ptrs(p_bnbwd[0])
        p2 = p_bnbwd[1].data_ptr()
    if q_act is not None:
        qm = 1
        qa, qb, _ = coef_ptrs(q_act[0])
        slope = q_act[1]
    check(lib.ms_conv_wgrad(p.data_ptr(), p2, q.data_ptr(), out.data_ptr(), N, M, Nq, Hp, Wp, Hq, Wq, ks, stride, q_fetch, pm, pa, pb, pc,
                            qm, qa, qb, 4, slope, 1 if accumulate else 0, ws.data_ptr(), ws.numel(), _stream()), "ms_conv_wgrad")
    return out


def mse_loss(x, target):
    """0.5 * mean((x - target)^2) as a 0-dim device tensor (ms_mse_loss)."""
    _need_cuda_f32(x); _need_cuda_f32(target)
    n = x.numel()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.ms_mse_ws_bytes(), dtype=torch.uint8, device=x.device)
    check(lib.ms_mse_loss(x.data_ptr(), target.data_ptr(), n, 0.5 / n, 0.0, out.data_ptr(), 0, ws.data_ptr(), ws.numel(), _stream()), "ms_mse_loss")
    return out[0]
