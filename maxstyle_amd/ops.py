"""Thin tensor-level wrappers over the C ABI (pointer/size marshaling only; PyTorch is used for device
memory and the current HIP stream, nothing else)."""
import torch

from ._lib import lib, check

_ws_cache = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _need_cuda_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("maxstyle_amd ops run on the MI355X only: got a CPU tensor (there is no CPU fallback)")
        if t.dtype != torch.float32:
            raise TypeError(f"maxstyle_amd ops are fp32; got {t.dtype}")
        if not t.is_contiguous():
            raise ValueError("maxstyle_amd ops need contiguous NCHW tensors")


def workspace(nbytes, device):
    """A cached scratch buffer per (device, stream); grown geometrically. Kernels on one stream are ordered, so reuse is safe."""
    key = (device, _stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def style_ws(B, C, HW, device):
    n = lib.ms_style_ws_bytes(B, C, HW)
    return workspace(n, device)


def style_moments(x, eps=1e-6):
    _need_cuda_f32(x)
    B, C = x.shape[:2]
    HW = x[0, 0].numel()
    mu = torch.empty(B, C, 1, 1, device=x.device, dtype=torch.float32)
    sig = torch.empty_like(mu)
    ws = style_ws(B, C, HW, x.device)
    check(lib.ms_style_moments(x.data_ptr(), mu.data_ptr(), sig.data_ptr(), B * C, HW, eps, ws.data_ptr(), ws.numel(), _stream()), "ms_style_moments")
    return mu, sig


def style_fwd(x, perm, lmda, gamma_noise, beta_noise, gamma_std, beta_std, compute_std, eps=1e-6, out=None):
    """Returns y, mu, sig, coefA, coefS. gamma_std/beta_std are [1,C,1,1] buffers (written if compute_std)."""
    _need_cuda_f32(x, lmda, gamma_noise, beta_noise, gamma_std, beta_std)
    B, C = x.shape[:2]
    HW = x[0, 0].numel()
    dev = x.device
    y = torch.empty_like(x) if out is None else out
    stats = torch.empty(4, B, C, 1, 1, device=dev, dtype=torch.float32)
    mu, sig, cA, cS = stats[0], stats[1], stats[2], stats[3]
    ws = style_ws(B, C, HW, dev)
    check(lib.ms_style_fwd(x.data_ptr(), y.data_ptr(), mu.data_ptr(), sig.data_ptr(), gamma_std.data_ptr(), beta_std.data_ptr(),
                           1 if compute_std else 0, _ptr(lmda), _ptr(gamma_noise), _ptr(beta_noise), _ptr(perm),
                           cA.data_ptr(), cS.data_ptr(), B, C, HW, eps, ws.data_ptr(), ws.numel(), _stream()), "ms_style_fwd")
    return y, mu, sig, cA, cS


def style_bwd(dy, x, mu, sig, coefA, gamma_std, beta_std, lmda, perm, need_dx, need_noise, need_lmda):
    _need_cuda_f32(dy, x)
    B, C = x.shape[:2]
    HW = x[0, 0].numel()
    dev = x.device
    dx = torch.empty_like(x) if need_dx else None
    dg = torch.empty(B, C, 1, 1, device=dev, dtype=torch.float32) if need_noise else None
    db = torch.empty(B, C, 1, 1, device=dev, dtype=torch.float32) if need_noise else None
    dl = torch.empty(B, 1, 1, 1, device=dev, dtype=torch.float32) if need_lmda else None
    ws = style_ws(B, C, HW, dev)
    check(lib.ms_style_bwd(dy.data_ptr(), x.data_ptr(), _ptr(dx), mu.data_ptr(), sig.data_ptr(), coefA.data_ptr(),
                           _ptr(gamma_std), _ptr(beta_std), _ptr(lmda), _ptr(perm), _ptr(dg), _ptr(db), _ptr(dl),
                           B, C, HW, ws.data_ptr(), ws.numel(), _stream()), "ms_style_bwd")
    return dx, dg, db, dl


def adam_step(p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-8, step_dev=None):
    _need_cuda_f32(p, g, m, v)
    check(lib.ms_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, b1, b2, eps, int(step),
                           _ptr(step_dev), _stream()), "ms_adam_step")
