"""Drop-in network modules for the MaxStyle hot path, backed by the HIP engine.

Mirrors the module / parameter layout of /root/reference/src/models/ebm/encoder_decoder.py so that reference
checkpoints (`state_dict` keys: SURVEY.md A.6) load unchanged:
    res_convdown (:22-74), res_up_family (:289-357), MyEncoder (:423-482), MyDecoder (+apply_max_style, :561-631),
    Dual_Branch_Encoder (:634-680), and the helper `_disable_tracking_bn_stats` (model_util.py:468-510).
The torch.nn layers below are *parameter containers only*: forward passes run on the MI355X through maxstyle_amd.engine
(hand-written HIP kernels).  Forward is inference/inner-loop only (no autograd graph through the network weights - inside
the inner loop they are frozen, advanced_triplet...py:508-511); gradients w.r.t. MaxStyle parameters flow through
`MyDecoder.apply_max_style` via a hand-written backward.  CPU tensors are refused (no fallback).
"""
import contextlib

import torch
import torch.nn as nn

from . import engine as E

LEAKY = 0.2


@contextlib.contextmanager
def _disable_tracking_bn_stats(model):
    """model_util.py:468-510: BN layers use batch statistics without touching running buffers; affine frozen."""
    old = {}
    for name, m in model.named_modules():
        if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
            old[name] = m.track_running_stats
            m.track_running_stats = False
            m.weight.requires_grad_(False)
            m.bias.requires_grad_(False)
    try:
        yield
    finally:
        for name, m in model.named_modules():
            if name in old:
                m.track_running_stats = old[name]
                m.weight.requires_grad_(old[name])
                m.bias.requires_grad_(old[name])


def module_params(module):
    """The module's parameters as a list, cached on the module: `generate_max_style_image` flips requires_grad on ~330 tensors six times per call and clears their
    gradients twice (advanced_triplet...py:508-511, 568-571), and nn.Module.parameters() re-walks the module tree every time (0.6 ms of host time per call, during
    which the GPU idles).  The Parameter OBJECTS of these containers never change (load_state_dict copies in place, the flat parameter bank re-points .data).
    The cache carries a cheap signature of the module tree - the number of sub-modules and of directly registered parameters over the whole tree - and is rebuilt
    when it changes (a later register_parameter / add_module / parametrize no longer escapes set_grad and zero_grad: ADVICE r3)."""
    ent = module.__dict__.get("_ms_plist")
    mods = module.__dict__.get("_ms_mlist")
    if ent is not None and mods is not None:
        sig = sum(len(m._parameters) + len(m._modules) for m in mods)
        if sig == ent[0]:
            return ent[1]
    mods = list(module.modules())
    pl = list(module.parameters())
    module.__dict__["_ms_mlist"] = mods
    module.__dict__["_ms_plist"] = (sum(len(m._parameters) + len(m._modules) for m in mods), pl)
    return pl


def set_grad(module, requires_grad=False):
    for p in module_params(module):
        p.requires_grad = requires_grad


def _double_conv(in_ch, out_ch, norm):
    return nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, padding=1, bias=True), norm(out_ch), nn.LeakyReLU(LEAKY),
                         nn.Conv2d(out_ch, out_ch, 3, padding=1, bias=True), norm(out_ch))


class res_convdown(nn.Module):
    def __init__(self, in_ch, out_ch, norm=nn.BatchNorm2d, if_SN=False, bias=True, dropout=None):
        super().__init__()
        if if_SN or dropout is not None:
            raise NotImplementedError("spectral norm / dropout variants are outside the MaxStyle hot path")
        self.down = nn.Conv2d(in_ch, in_ch, 3, stride=2, padding=1, bias=bias)
        self.conv = _double_conv(in_ch, out_ch, norm)
        self.conv_input = nn.Conv2d(in_ch, out_ch, kernel_size=1, stride=1, padding=0, bias=bias)
        self.last_act = nn.LeakyReLU(LEAKY)
        self.dropout = None


class res_up_family(nn.Module):
    def __init__(self, in_ch, out_ch, norm=nn.BatchNorm2d, if_SN=False, bias=True, dropout=None, up_type='Conv2'):
        super().__init__()
        if if_SN or dropout is not None:
            raise NotImplementedError("spectral norm / dropout variants are outside the MaxStyle hot path")
        if up_type == 'NN':
            self.up = nn.Sequential(nn.UpsamplingNearest2d(scale_factor=2))
        elif up_type == 'Conv2':
            self.up = nn.ConvTranspose2d(in_ch, in_ch, kernel_size=2, stride=2)
        else:
            raise NotImplementedError(up_type)
        self.up_type = up_type
        self.conv = _double_conv(in_ch, out_ch, norm)
        self.conv_input = nn.Conv2d(in_ch, out_ch, kernel_size=1, stride=1, padding=0, bias=bias)
        self.last_act = nn.LeakyReLU(LEAKY)
        self.dropout = None


class _HipNet(nn.Module):
    """Shared plumbing: cached packed weights + an engine per input shape."""

    _weights_epoch = 0      # bumped by writers that bypass torch (the flat optimiser's HIP kernel, raw-pointer running-statistics updates)

    def _sd_key(self):
        """Identity of the weights the packed copies were built from: torch-side writes bump `_version`; the flat-buffer optimiser
        (solver._BankOptimizer.step -> ms_adamw_step) writes through raw pointers and bumps `_weights_epoch` instead."""
        return (self._weights_epoch,) + tuple((id(p), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def note_weights_changed(self):
        self._weights_epoch += 1

    def _engine(self, spec, B, H, W, dev, pack):
        key = self._sd_key()
        if getattr(self, "_packed_key", None) != key or self._packed is None:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            self._packed = pack(sd)
            self._packed_key = key
            self._engines = {}
        ek = (B, H, W, str(dev))
        eng = self._engines.get(ek)
        if eng is None:
            eng = E.InnerLoopEngine(spec, B, H, W, dev)
            self._engines[ek] = eng
        return eng, self._packed

    @staticmethod
    def _check(x):
        if not x.is_cuda:
            raise RuntimeError("maxstyle_amd networks run on the MI355X only (HIP kernels); got a CPU tensor")
        return x.contiguous().float()

    def _bn_flags(self):
        """(use running statistics?, update running statistics?) from module.training / track_running_stats (all BN layers agree)."""
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        tracking = all(m.track_running_stats for m in bns)
        return (not self.training), (self.training and tracking)

    def _observer(self, enabled):
        if not enabled:
            return None
        by_name = {n: m for n, m in self.named_modules() if isinstance(m, nn.BatchNorm2d)}

        def obs(bnw, coef):
            m = by_name[bnw.name]
            mean, invstd = coef[:, 2], coef[:, 3]
            n = float(self._bn_count[bnw.name]) if hasattr(self, "_bn_count") and bnw.name in self._bn_count else None
            var = 1.0 / (invstd * invstd) - E.BN_EPS
            if n is not None and n > 1:
                var = var * (n / (n - 1.0))          # running_var tracks the unbiased estimate
            mom = m.momentum if m.momentum is not None else 0.1
            with torch.no_grad():
                m.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                m.running_var.mul_(1 - mom).add_(var, alpha=mom)
                m.num_batches_tracked += 1
        return obs


class MyEncoder(_HipNet):
    def __init__(self, input_channel, output_channel=None, feature_reduce=1, encoder_dropout=None, norm=nn.BatchNorm2d, if_SN=False, act=None):
        super().__init__()
        r = feature_reduce
        self.inc = nn.Sequential(nn.Conv2d(input_channel, 64 // r, 3, padding=1, bias=True), norm(64 // r), nn.LeakyReLU(LEAKY),
                                 nn.Conv2d(64 // r, 64 // r, 3, padding=1, bias=True), norm(64 // r))
        self.down1 = res_convdown(64 // r, 128 // r, norm=norm)
        self.down2 = res_convdown(128 // r, 256 // r, norm=norm)
        self.down3 = res_convdown(256 // r, 512 // r, norm=norm)
        self.down4 = res_convdown(512 // r, 512 // r, norm=norm)
        out = 512 // r if output_channel is None else output_channel
        self.final_conv = nn.Sequential(nn.Conv2d(512 // r, out, kernel_size=1, stride=1, padding=0), norm(out))
        self.act = act if act is not None else nn.ReLU()
        self.feature_reduce = r
        self.input_channel = input_channel


class Dual_Branch_Encoder(_HipNet):
    """FTN encoder: z_i = ReLU(BN(conv1x1(down4(...)))), z_s = code_decoupler(z_i)  (encoder_decoder.py:634-680)."""

    def __init__(self, input_channel, z_level_1_channel=None, z_level_2_channel=None, feature_reduce=1, encoder_dropout=None,
                 norm=nn.BatchNorm2d, if_SN=False, num_domains=1):
        super().__init__()
        if num_domains > 1 or if_SN or encoder_dropout is not None:
            raise NotImplementedError("domain-specific BN / spectral norm / dropout encoders are outside the MaxStyle hot path")
        self.general_encoder = MyEncoder(input_channel, output_channel=z_level_1_channel, feature_reduce=feature_reduce, norm=norm, act=nn.ReLU())
        self.code_decoupler = nn.Sequential(
            nn.Conv2d(z_level_1_channel, z_level_2_channel, 3, padding=1, bias=False), norm(z_level_2_channel), nn.LeakyReLU(LEAKY),
            nn.Conv2d(z_level_2_channel, z_level_2_channel, 3, padding=1, bias=False), norm(z_level_2_channel), nn.ReLU())
        self.spec = E.NetSpec(feature_reduce, input_channel, 0)
        self._packed = None

    def _run(self, x):
        x = self._check(x)
        B, _, H, W = x.shape
        eng, nets = self._engine(self.spec, B, H, W, x.device, lambda sd: E.PackedNets(self.spec, enc_sd=sd))
        eng.nets = nets
        eng.bn_eval, track = self._bn_flags()
        eng.bn_observer = self._observer(track)
        self._bn_count = _ElemCounts(B, H, W)
        z_i, z_s = eng.encode_fwd(x)
        eng.bn_observer = None
        return z_i, z_s

    def forward(self, x, domain_id=0):
        z_i, z_s = self._run(x)
        return z_i.clone(), z_s.clone()

    def encode(self, x):
        """general_encoder(x) only (z_i)."""
        return self._run(x)[0].clone()

    def filter_code(self, z):
        """encoder_decoder.py:673-675: z_s = code_decoupler(z) for a code z that did not come out of this module's own forward."""
        z = self._check(z)
        B, _, h, w = z.shape
        eng, nets = self._engine(self.spec, B, h * 16, w * 16, z.device, lambda sd: E.PackedNets(self.spec, enc_sd=sd))
        eng.nets = nets
        eng.bn_eval, track = self._bn_flags()
        eng.bn_observer = self._observer(track)
        self._bn_count = _ElemCounts(B, h * 16, w * 16)
        z_s = eng.decouple_fwd(z)
        eng.bn_observer = None
        return z_s.clone()


class _ElemCounts(dict):
    """Number of elements per channel seen by each BatchNorm of the encoder (for the unbiased running_var update)."""

    def __init__(self, B, H, W):
        super().__init__()
        g = "general_encoder."
        self[g + "inc.1"] = self[g + "inc.4"] = B * H * W
        for i in range(1, 5):
            n = B * (H >> i) * (W >> i)
            self[g + f"down{i}.conv.1"] = self[g + f"down{i}.conv.4"] = n
        n = B * (H >> 4) * (W >> 4)
        self[g + "final_conv.1"] = self["code_decoupler.1"] = self["code_decoupler.4"] = n


class MyDecoder(_HipNet):
    def __init__(self, input_channel, output_channel, feature_reduce=1, decoder_dropout=None, norm=nn.BatchNorm2d, up_type='Conv2', if_SN=False, last_act=None):
        super().__init__()
        if if_SN or decoder_dropout is not None:
            raise NotImplementedError("spectral norm / dropout decoders are outside the MaxStyle hot path")
        r = feature_reduce
        self.up1 = res_up_family(input_channel, 256 // r, norm=norm, up_type=up_type)
        self.up2 = res_up_family(256 // r, 128 // r, norm=norm, up_type=up_type)
        self.up3 = res_up_family(128 // r, 64 // r, norm=norm, up_type=up_type)
        self.up4 = res_up_family(64 // r, 64 // r, norm=norm, up_type=up_type)
        self.final_conv = nn.Conv2d(64 // r, output_channel, kernel_size=1, stride=1, padding=0)
        self.last_act = last_act
        self.up_type = up_type
        self.spec = E.NetSpec(r, output_channel, output_channel)
        self._packed = None
        if last_act is not None and not isinstance(last_act, nn.Sigmoid):
            raise NotImplementedError("only nn.Sigmoid (intensity_norm_type='min_max') or None as last_act")

    def _bn_counts(self, B, h, w):
        d = {}
        for i in range(1, 5):
            d[f"up{i}.conv.1"] = d[f"up{i}.conv.4"] = B * (h << i) * (w << i)
        return d

    def _prep(self, code, force_batch_stats=False):
        code = self._check(code)
        B, _, h, w = code.shape
        pack = (lambda sd: E.PackedNets(self.spec, dec_sd=sd)) if self.up_type == 'Conv2' else (lambda sd: E.PackedNets(self.spec, seg_sd=sd))
        eng, nets = self._engine(self.spec, B, h * 16, w * 16, code.device, pack)
        eng.nets = nets
        eval_bn, track = self._bn_flags()
        eng.bn_eval = eval_bn and not force_batch_stats
        eng.bn_observer = self._observer(track and not force_batch_stats)
        self._bn_count = self._bn_counts(B, h, w)
        return eng, code

    def forward(self, x):
        eng, code = self._prep(x)
        if self.up_type == 'Conv2':
            eng.configure_styles([], {})
            eng._prefix_valid = False
            out = eng.decode(code)
            if self.last_act is None:
                raise NotImplementedError("ConvTranspose decoder without Sigmoid is not on the MaxStyle path")
        else:
            if self.last_act is not None:
                raise NotImplementedError("NN-upsampling decoder with a last activation is not on the MaxStyle path")
            out = eng.seg_logits(code)
        eng.bn_observer = None
        return out.clone()

    def apply_max_style(self, image_code, nn_style_augmentor_dict, decoder_layers_indexes=[3, 4, 5]):
        """encoder_decoder.py:598-631.  BN runs on batch statistics (every block is wrapped in _disable_tracking_bn_stats there),
        the code is detached; the result is differentiable w.r.t. the MaxStyle parameters in `nn_style_augmentor_dict`."""
        if self.up_type != 'Conv2' or self.last_act is None:
            raise NotImplementedError("apply_max_style is implemented for the image decoder (up_type='Conv2', Sigmoid)")
        eng, code = self._prep(image_code.detach(), force_batch_stats=self.training)
        mods = {int(k): m for k, m in nn_style_augmentor_dict.items() if int(k) in decoder_layers_indexes}
        return _ApplyMaxStyle.run(eng, code, mods, sorted(mods))


class _ApplyMaxStyle(torch.autograd.Function):
    @staticmethod
    def run(eng, code, mods, layers):
        slots = E.slots_from_modules(mods, code.device)
        applied = [i for i in layers if i in slots]
        eng.configure_styles(applied, slots)
        params = []
        for i in applied:
            m = mods[i]
            eng.set_style_state(i, m.perm, m.lmda.detach(), m.gamma_noise.detach(), m.beta_noise.detach())
            if m.gamma_std is not None and m.beta_std is not None:       # statistics frozen by an earlier forward
                std = eng.t(f"st{i}.std", 2, slots[i].C)
                std[0].copy_(m.gamma_std.reshape(-1)); std[1].copy_(m.beta_std.reshape(-1))
                slots[i].have_std = True
            params += [m.gamma_noise, m.beta_noise, m.lmda]
        return _ApplyMaxStyle.apply(eng, code, mods, applied, *params)

    @staticmethod
    def forward(ctx, eng, code, mods, applied, *params):
        eng._prefix_valid = False
        eng.code = code
        out = eng.decode(code).clone()
        for i in applied:                                                # publish the (now frozen) batch statistics
            std = eng.buf.get(f"st{i}.std")
            if std is not None and mods[i].gamma_std is None:
                C = std.shape[1]
                mods[i].gamma_std = std[0].clone().view(1, C, 1, 1)
                mods[i].beta_std = std[1].clone().view(1, C, 1, 1)
            mods[i].data = None
        ctx.eng, ctx.applied = eng, applied
        return out

    @staticmethod
    def backward(ctx, dout):
        eng, applied = ctx.eng, ctx.applied
        if not applied:
            return (None,) * 4
        eng.flat_g.zero_()
        eng.decode_bwd(dout.contiguous().clone())
        grads = []
        for i in applied:
            for nm in ("gamma_noise", "beta_noise", "lmda"):
                grads.append(eng.grad(i, nm).clone())
        return (None, None, None, None, *grads)
