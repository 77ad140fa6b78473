"""`MaxStyle` nn.Module - drop-in for /root/reference/src/advanced/maxstyle.py:6-189 backed by the fused HIP
kernels (K1 forward / K2 backward).  Same constructor, attributes, parameter order (gamma_noise, beta_noise,
lmda), RNG draws (CPU generator for perm/rand_p, device generator for the noise / lmda), identity short-cuts
and assertion messages.  GPU only: a CPU tensor raises (no fallback by design; the CPU restatement used by the
tests lives in oracle/ and is never imported here)."""
import torch
import torch.nn as nn

from . import ops


class _MaxStyleFunction(torch.autograd.Function):
    """y = A((x-mu)/sig)+S with mu/sig treated as constants (maxstyle.py:160 detaches them)."""

    @staticmethod
    def forward(ctx, x, gamma_noise, beta_noise, lmda, layer):
        xc = x.contiguous()
        use_mix = layer.mix_style
        use_noise = not layer.no_noise
        compute_std = layer.gamma_std is None or layer.beta_std is None
        C = xc.shape[1]
        if layer.gamma_std is None:
            layer.gamma_std = torch.empty(1, C, 1, 1, device=xc.device, dtype=torch.float32)
        if layer.beta_std is None:
            layer.beta_std = torch.empty(1, C, 1, 1, device=xc.device, dtype=torch.float32)
        perm = layer._perm_device(xc.device)
        y, mu, sig, cA, cS = ops.style_fwd(
            xc, perm if use_mix else None, lmda.detach().contiguous() if use_mix else None,
            gamma_noise.detach().contiguous() if use_noise else None, beta_noise.detach().contiguous() if use_noise else None,
            layer.gamma_std, layer.beta_std, compute_std, layer.eps)
        ctx.save_for_backward(xc, mu, sig, cA, lmda.detach())
        ctx.layer = layer
        ctx.use_mix, ctx.use_noise = use_mix, use_noise
        layer._last_stats = (mu, sig)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mu, sig, cA, lmda = ctx.saved_tensors
        layer = ctx.layer
        need_dx, need_g, need_b, need_l = ctx.needs_input_grad[:4]
        need_noise = (need_g or need_b) and ctx.use_noise
        need_l = need_l and ctx.use_mix
        perm = layer._perm_device(x.device)
        dx, dg, db, dl = ops.style_bwd(dy.contiguous(), x, mu, sig, cA, layer.gamma_std, layer.beta_std,
                                       lmda.contiguous() if ctx.use_mix else None, perm if ctx.use_mix else None,
                                       need_dx, need_noise, need_l)
        return dx, (dg if need_g else None), (db if need_b else None), (dl if need_l else None), None


_CUDA, _CPU = torch.device('cuda'), torch.device('cpu')


class MaxStyle(nn.Module):
    """MaxStyle layer (Chen et al., MICCAI 2022). See the reference docstring for argument meaning
    (/root/reference/src/advanced/maxstyle.py:14-30)."""

    def __init__(self, batch_size, num_feature, p=0.5, mix_style=True, no_noise=False,
                 mix_learnable=True, noise_learnable=True, always_use_beta=False, alpha=0.1, eps=1e-6, use_gpu=True, debug=False):
        super().__init__()
        # plain (non-tensor) attributes: straight into __dict__ - nn.Module.__setattr__ walks its parameter / buffer / module tables for every assignment, and the
        # trainer builds three of these layers per generate_max_style_image call (host time of the call: tools/prof_call_host.py)
        self.__dict__.update(batch_size=batch_size, num_feature=num_feature, p=p, mix_style=mix_style, no_noise=no_noise, mix_learnable=mix_learnable,
                             noise_learnable=noise_learnable, always_use_beta=always_use_beta, alpha=alpha, eps=eps, use_gpu=use_gpu, debug=debug,
                             device=_CUDA if use_gpu else _CPU, data=None, _perm_cache=None, _last_stats=None)
        if batch_size <= 1:
            # the reference spins forever here (identity-permutation rejection loop, maxstyle.py:55-58)
            raise ValueError("MaxStyle needs batch_size >= 2 (a batch of one has no non-identity permutation)")
        self.init_parameters()

    def init_parameters(self):
        """perm / rand_p from the CPU generator, noise / lmda from the device generator (maxstyle.py:48-122)."""
        B, C = self.batch_size, self.num_feature
        perm = torch.randperm(B)
        while torch.equal(perm, torch.arange(B)):
            perm = torch.randperm(B)
        self.__dict__.update(perm=perm, _perm_cache=None, rand_p=torch.rand(1))
        if self.rand_p >= self.p:
            for n in ("gamma_noise", "beta_noise", "lmda"):
                if n in self._parameters:
                    # the reference raises TypeError here (plain tensor over a registered Parameter, maxstyle.py:67)
                    raise TypeError(f"cannot assign 'torch.FloatTensor' as parameter '{n}' (torch.nn.Parameter or None expected)")
            self.gamma_noise = torch.zeros(B, C, 1, 1, device=self.device).float()
            self.beta_noise = torch.zeros(B, C, 1, 1, device=self.device).float()
            self.lmda = torch.zeros(B, 1, 1, 1, device=self.device).float()
        else:
            if self.noise_learnable:
                assert self.no_noise is False, 'turn no_noise=False to enable the optimization of noise'
                if self.device.type == "cuda":
                    # both N(0,1) tensors from ONE draw of the device generator (the reference draws them one after the other from its CUDA generator, whose stream this
                    # hardware cannot reproduce anyway: the same distribution, one launch instead of two per layer of a generate_max_style_image call)
                    gb = torch.empty(2, B, C, 1, 1, device=self.device).normal_()
                    self.gamma_noise = nn.Parameter(gb[0])
                    self.beta_noise = nn.Parameter(gb[1])
                else:
                    # CPU generator: the reference's draw order, number for number (tests/golden/kat_ramp.npz)
                    self.gamma_noise = nn.Parameter(torch.empty(B, C, 1, 1, device=self.device))
                    self.beta_noise = nn.Parameter(torch.empty(B, C, 1, 1, device=self.device))
                    nn.init.normal_(self.gamma_noise)
                    nn.init.normal_(self.beta_noise)
            else:
                # NB (reference quirk kept): fixed noise is N(0,1) only when no_noise=True, zeros otherwise (maxstyle.py:75-80)
                mk = torch.randn if self.no_noise else torch.zeros
                self._set_plain("gamma_noise", mk(B, C, 1, 1, device=self.device).float())
                self._set_plain("beta_noise", mk(B, C, 1, 1, device=self.device).float())
            if self.mix_style is False:
                self._set_plain("lmda", torch.zeros(B, 1, 1, 1, dtype=torch.float32, device=self.device))
            else:
                if self.always_use_beta:
                    self.beta_sampler = torch.distributions.Beta(self.alpha, self.alpha)
                    lmda = self.beta_sampler.sample((B, 1, 1, 1)).to(self.device)
                else:
                    lmda = torch.rand(B, 1, 1, 1, dtype=torch.float32, device=self.device)
                self.lmda = nn.Parameter(lmda.float())
                self.lmda.requires_grad = bool(self.mix_learnable)
        self.__dict__.update(gamma_std=None, beta_std=None)
        if self.debug:
            print("lmda:", self.lmda); print("gamma_noise:", self.gamma_noise); print("beta_noise:", self.beta_noise); print("perm:", self.perm)

    def _set_plain(self, name, value):
        if name in self._parameters:
            raise TypeError(f"cannot assign 'torch.FloatTensor' as parameter '{name}' (torch.nn.Parameter or None expected)")
        object.__setattr__(self, name, value)

    def _perm_device(self, device):
        if self._perm_cache is None or self._perm_cache[0] is not self.perm or self._perm_cache[1].device != device:
            self._perm_cache = (self.perm, self.perm.to(device=device, dtype=torch.int64).contiguous())
        return self._perm_cache[1]

    def __repr__(self):
        if self.p >= self.rand_p:
            return f'MaxStyle: \
                 mean of gamma noise: {torch.mean(self.gamma_noise)}, std:{torch.std(self.gamma_noise)} ,\
                 mean of beta noise: {torch.mean(self.beta_noise)}, std: {torch.std(self.beta_noise)}, \
                 mean of mix coefficient: {torch.mean(self.lmda)}, std: {torch.std(self.lmda)}'
        return 'diffuse style not applied'

    def reset(self):
        self.init_parameters()
        if self.debug:
            print('reinitializing parameters')

    def is_identity(self, x):
        B = x.size(0)
        hw = x.view(B, x.size(1), -1).size(2)
        return bool(self.rand_p >= self.p) or (not self.mix_style and self.no_noise) or B <= 1 or hw == 1

    def forward(self, x):
        self.data = x
        if self.is_identity(x):
            return x
        B, C = x.size(0), x.size(1)
        assert self.batch_size == B and self.num_feature == C, f"check input dim, expect ({self.batch_size}, {self.num_feature}, *,*) , got {B}{C}"
        if not x.is_cuda:
            raise RuntimeError("maxstyle_amd.MaxStyle runs on the MI355X only (HIP kernels); got a CPU tensor")
        return _MaxStyleFunction.apply(x, self.gamma_noise, self.beta_noise, self.lmda, self)
