"""`MixStyle` (and its DSU / 'gaussian' mode) - drop-in for /root/reference/src/advanced/mixstyle.py:6-108 on the MaxStyle HIP kernels.

The arithmetic is the MaxStyle layer's without learnable parameters: per-(b,c) moments, style statistics mixed with a
permuted sample (`random` / `crossdomain`) with an UN-clamped lmda, or perturbed with N(0,1) x batch-std noise (`gaussian`, DSU);
mu / sig are detached, so the backward is dx = dy * A / sig only.  Same RNG draw order as the reference (`torch.rand(1)`, Beta sample,
`randperm` on the CPU generator; the DSU noise on the input's device).  GPU tensors only."""
import torch
import torch.nn as nn

from . import ops


class _MixStyleFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, perm, lmda, gnoise, bnoise, eps):
        xc = x.contiguous()
        C = xc.shape[1]
        gs = torch.empty(1, C, 1, 1, device=xc.device, dtype=torch.float32)
        bs = torch.empty_like(gs)
        flags = 2 | (1 if gnoise is not None else 0)          # bit 1: lmda is not clamped; bit 0: batch std recomputed on every call (DSU)
        y, mu, sig, cA, cS = ops.style_fwd(xc, perm, lmda, gnoise, bnoise, gs, bs, flags, eps)
        ctx.save_for_backward(xc, mu, sig, cA)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mu, sig, cA = ctx.saved_tensors
        dx, _, _, _ = ops.style_bwd(dy.contiguous(), x, mu, sig, cA, None, None, None, None, True, False, False)
        return dx, None, None, None, None, None


class MixStyle(nn.Module):
    def __init__(self, p=0.5, alpha=0.1, eps=1e-8, mix='random', lmda=None, zero_init=False, coefficient_sampler=None):
        super().__init__()
        self.p = p
        self.eps = eps
        self.mu = None
        self.std = None
        self.zero_init = zero_init
        self.alpha = alpha
        self.mix = mix
        self._activated = True
        self.lmda = lmda
        self.coeficient_sampler = None          # (sic) the reference never stores its constructor argument: Beta is always used
        self.beta = torch.distributions.Beta(alpha, alpha)

    def __repr__(self):
        return f'MixStyle(p={self.p}, alpha={self.alpha}, eps={self.eps}, mix={self.mix})'

    def update_mix_method(self, mix='random'):
        self.mix = mix

    def get_perm(self):
        return self.perm

    def forward(self, x, perm=None):
        p = torch.rand(1)
        if p > self.p:
            return x
        if not x.is_cuda:
            raise RuntimeError("maxstyle_amd.MixStyle runs on the MI355X only (HIP kernels); got a CPU tensor")
        B, C = x.size(0), x.size(1)
        if self.lmda is None:
            lmda = self.beta.sample((B, 1, 1, 1))
        else:
            lmda = torch.ones(B, 1, 1, 1) * self.lmda
        lmda = lmda.to(device=x.device, dtype=torch.float32).contiguous()
        if self.mix in ['random', 'crossdomain']:
            if perm is None:
                if self.mix == 'random':
                    perm = torch.randperm(B)
                else:
                    perm = torch.arange(B - 1, -1, -1)
                    perm_b, perm_a = perm.chunk(2)
                    perm_b = perm_b[torch.randperm(B // 2)]
                    perm_a = perm_a[torch.randperm(B // 2)]
                    perm = torch.cat([perm_b, perm_a], 0)
            self.perm = perm
            return _MixStyleFunction.apply(x, perm.to(device=x.device, dtype=torch.int64).contiguous(), lmda, None, None, self.eps)
        elif self.mix == 'gaussian':
            # DSU: mu + N(0,1)*std_b(mu), sig + N(0,1)*std_b(sig); the MaxStyle kernel's gamma_noise multiplies std_b(sig), beta_noise std_b(mu)
            gaussian_mu = torch.randn(B, C, 1, 1, device=x.device)
            gaussian_std = torch.randn(B, C, 1, 1, device=x.device)
            inj = getattr(self, "_inject_noise", None)       # test seam (device RNG streams differ between CPU and GPU)
            if inj is not None:
                gaussian_mu, gaussian_std = inj[0].to(x.device).float(), inj[1].to(x.device).float()
            return _MixStyleFunction.apply(x, None, None, gaussian_std.contiguous(), gaussian_mu.contiguous(), self.eps)
        raise NotImplementedError
