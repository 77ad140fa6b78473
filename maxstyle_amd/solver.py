"""`AdvancedTripletReconSegmentationModel` - the hot-path surface of the reference solver
(/root/reference/src/models/advanced_triplet_recon_segmentation_model.py) on the MI355X engine.

Implemented (same names, argument meaning and error behaviour):
    __init__/get_network (:42-266, FCN_16 / FCN_64 families), encode_image / filter_code (:330-385),
    decoder_inference (:693-716), generate_max_style_image (:458-571), predict-style evaluate helper, zero_grad/train/eval.
The K-step inner loop runs in maxstyle_amd.engine (HIP kernels + HIP-graph replay); this file is host-side orchestration.
Outer update (SURVEY.md 8(f) rows 1,3): standard_training (:731-786), hard_example_traininng (:843-889), fast_predict (:891-912),
    compute_image_recon_loss 'l2' (:718-729), set_optimizers / reset_all_optimizers / optimize_all_params / optimize_params /
    reset_optimizer (:1038-1091) - forward AND backward (weight gradients) in maxstyle_amd.train_engine, optimiser on one flat buffer.
Out of scope here (SURVEY.md 8 OUT): the shape-refinement (STN) networks, other augmentation baselines, checkpoint I/O helpers.
"""
import os
import torch
import torch.nn as nn

from . import engine as E
from . import train_engine as T
from .maxstyle import MaxStyle
from .networks import Dual_Branch_Encoder, MyDecoder, _disable_tracking_bn_stats, set_grad, module_params


def cross_entropy_2D(input, target, weight=None, size_average=True):
    """custom_loss.py:1043-1078 for 3-D label targets: sum of pixel NLL / (N*H*W). HIP kernel (fused log-softmax + NLL)."""
    from . import ops
    if weight is not None or not size_average or target.dim() != 3:
        raise NotImplementedError("only the un-weighted, size-averaged label-map form used by the MaxStyle loop")
    n, c, h, w = input.shape
    eye = torch.eye(c, device=input.device, dtype=torch.float32)
    loss, _, _ = ops.head_ce(input.contiguous().float(), eye, None, target.contiguous(), loss_sign=1.0, need_dh=False)
    return loss[0]


def basic_loss_fn(pred, target, loss_type='cross_entropy', class_weights=None, use_gpu=True):
    """custom_loss.py:13-45 ('cross entropy' branch - the only one the MaxStyle loop uses; class_weights are ignored by it)."""
    if class_weights is not None:
        assert len(class_weights) == pred.size(1), 'each cls must have a weight, expect to have {} classses but got {} weights'.format(pred.size(1), len(class_weights))
    if loss_type == 'cross entropy':
        return cross_entropy_2D(pred, target)
    raise NotImplementedError


class AdvancedTripletReconSegmentationModel(nn.Module):
    def __init__(self, network_type='FCN_16_standard', image_ch=1, learning_rate=1e-4, encoder_dropout=None, decoder_dropout=None,
                 num_classes=4, n_iter=1, checkpoint_dir=None, use_gpu=True, debug=False, rec_loss_type='l2', separate_training=False,
                 class_weights=None, optimizer_type='Adam', image_size=192, intensity_norm_type="min_max"):
        super().__init__()
        self.network_type = network_type
        self.image_ch = image_ch
        self.learning_rate = learning_rate
        self.num_classes = num_classes
        self.n_iter = n_iter
        self.checkpoint_dir = checkpoint_dir
        self.use_gpu = use_gpu
        self.debug = debug
        self.class_weights = class_weights
        self.intensity_norm_type = intensity_norm_type
        self.image_size = image_size
        if encoder_dropout is not None or decoder_dropout is not None:
            raise NotImplementedError("dropout is None in every MaxStyle config (SURVEY.md 2 row 6)")
        self.latent_code = {}
        if rec_loss_type != 'l2':
            raise NotImplementedError("rec_loss_type 'l2' (0.5*MSE) is what every MaxStyle config uses")
        if optimizer_type not in ('Adam', 'AdamW'):
            raise NotImplementedError("optimizer_type 'Adam' or 'AdamW'")
        self.rec_loss_type = rec_loss_type
        self.optimizer_type = optimizer_type
        self.separate_training = separate_training
        self.optimizers = None
        self.z_i = self.z_s = None
        self.recon_image = None
        self._bank = None
        self._train_engines = {}
        self._anchor = None
        self.model = self.get_network(checkpoint_dir=checkpoint_dir)
        self._engines = {}
        self._packed = None
        self._packed_key = None
        self.last_style_modules = None
        self.schedulers_dict = {}            # advanced_triplet...py:1070-1081: schedulers exist for SGD only ("No schedulers for optimizers")
        self.running_metric = self.set_running_metric()       # advanced_triplet...py:93 (the trainer calls .reset() before the first evaluate(), train_adv...py:77)
        self.cur_eval_images = self.cur_eval_predicts = self.cur_eval_gts = None

    # ------------------------------------------------------------------ construction (advanced_triplet...py:125-266)
    def get_network(self, checkpoint_dir=None):
        nt = self.network_type
        if self.intensity_norm_type != 'min_max' or 'z_score' in nt or 'identity' in nt:
            raise NotImplementedError("only the Sigmoid image decoder (intensity_norm_type='min_max') is on the MaxStyle path")
        if nt.startswith('Unet') or 'DS_FCN' in nt or not ('FCN_16' in nt or 'FCN_64' in nt):
            print(f'no {nt} found')
            raise NotImplementedError
        reduce_factor = 4 if '16' in nt else 1
        self.reduce_factor = reduce_factor
        image_encoder = Dual_Branch_Encoder(input_channel=self.image_ch, z_level_1_channel=512 // reduce_factor,
                                            z_level_2_channel=512 // reduce_factor, feature_reduce=reduce_factor, norm=nn.BatchNorm2d, num_domains=1)
        segmentation_decoder = MyDecoder(input_channel=512 // reduce_factor, up_type='NN', output_channel=self.num_classes,
                                         feature_reduce=reduce_factor, norm=nn.BatchNorm2d)
        model = {'image_encoder': image_encoder, 'segmentation_decoder': segmentation_decoder}
        if 'no_im_recon' not in nt:
            up = 'NN' if 'NN_decoder' in nt else 'Conv2'
            model['image_decoder'] = MyDecoder(input_channel=512 // reduce_factor, up_type=up, output_channel=self.image_ch,
                                               feature_reduce=reduce_factor, norm=nn.BatchNorm2d, last_act=nn.Sigmoid())
        for name, module in model.items():
            self._init_weights(module)
            if checkpoint_dir:
                import os
                path = os.path.join(checkpoint_dir, f'{name}.pth')
                if os.path.exists(path):
                    module.load_state_dict(torch.load(path, map_location='cpu'))
            if self.use_gpu:
                module.cuda()
            self.add_module(name, module)          # so .parameters()/.to()/.train() see the three sub-nets
        return model

    @staticmethod
    def _init_weights(module):
        """init_weight.py:52-61 semantics: kaiming conv weights, BN gamma ~ N(1,0.02), beta 0; ConvTranspose keeps the default."""
        for m in module.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.normal_(m.weight.data, 1.0, 0.02)
                nn.init.constant_(m.bias.data, 0.0)

    def get_modules(self):
        return self.model.values

    def zero_grad(self, set_to_none=True):
        if self._bank is not None:
            self._bank.zero_grad()              # the gradients are views of one flat buffer: keep them attached, clear the values
            return
        for m in self.model.values():
            if set_to_none:
                for p in module_params(m):          # nn.Module.zero_grad(set_to_none=True) on the cached parameter list (see networks.module_params)
                    p.grad = None
            else:
                m.zero_grad(set_to_none=False)

    def train(self, mode=True, if_testing=False):
        """nn.Module.train(mode); `if_testing=True` is the reference's spelling of eval (advanced_triplet...py:1018-1035)."""
        if if_testing:
            mode = False
        for m in self.model.values():
            m.train(mode)
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    # ------------------------------------------------------------------ thin dispatch (advanced_triplet...py:330-385, 693-716)
    def encode_image(self, input, domain_id=0, disable_track_bn_stats=False):
        enc = self.model['image_encoder']
        if disable_track_bn_stats:
            with _disable_tracking_bn_stats(enc):
                z_i, z_s = enc(input)
        else:
            z_i, z_s = enc(input)
        if 'w_o_filter' in self.network_type:
            z_s = z_i
        if 'share_code' in self.network_type:
            z_i = z_s
        self.latent_code['image'] = z_i
        self.latent_code['segmentation'] = z_s
        return z_i, z_s

    def filter_code(self, z, disable_track_bn_stats=False):
        """advanced_triplet...py:347-385 for the FCN families: latent_code_i = z, latent_code_s = image_encoder.filter_code(z)
        ('w_o_filter': both are z; 'share_code': both are the filtered code)."""
        if self.network_type.startswith('Unet'):
            raise NotImplementedError("UNet backbones are outside the MaxStyle path (SURVEY.md 2)")
        latent_code_i = z
        if 'w_o_filter' in self.network_type:
            latent_code_s = z
        else:
            enc = self.model['image_encoder']
            if disable_track_bn_stats:
                with _disable_tracking_bn_stats(enc):
                    latent_code_s = enc.filter_code(z)
            else:
                latent_code_s = enc.filter_code(z)
            if 'share_code' in self.network_type:
                latent_code_i = latent_code_s
        self.latent_code['image'] = latent_code_i
        self.latent_code['segmentation'] = latent_code_s
        return latent_code_i, latent_code_s

    def decoder_inference(self, latent_code, decoder_name="", decoder=None, eval=False, disable_track_bn_stats=False):
        if decoder is None:
            try:
                decoder = self.model[decoder_name]
            except Exception:
                raise ValueError("error! no decoder named {}".format(decoder_name))
        decoder_state = decoder.training
        if eval:
            decoder.eval()
            with torch.no_grad():
                logit = decoder(latent_code)
        elif disable_track_bn_stats:
            with _disable_tracking_bn_stats(decoder):
                logit = decoder(latent_code)
        else:
            logit = decoder(latent_code)
        decoder.train(mode=decoder_state)
        return logit

    def run(self, input, disable_track_bn_stats=False, normalize_input=False):
        """advanced_triplet...py:310-328 for 'no_STN' networks: (recon_image, init_predict, refined_predict = init_predict); side effect z_i / z_s."""
        from . import ops
        if normalize_input:
            if self.intensity_norm_type != 'min_max':
                raise NotImplementedError
            input = ops.rescale_intensity(input.detach().contiguous().float(), 0.0, 1.0)
        (z_i, z_s), init_predict = self.fast_predict(input, disable_track_bn_stats=disable_track_bn_stats)
        self.z_i, self.z_s = z_i, z_s
        recon_image = self.decoder_inference(decoder_name='image_decoder', latent_code=z_i, disable_track_bn_stats=disable_track_bn_stats)
        if 'no_STN' not in self.network_type:
            raise NotImplementedError("shape-refinement (STN) networks are outside the MaxStyle path (SURVEY.md 8)")
        return recon_image, init_predict, init_predict

    def predict(self, input, softmax=False, n_iter=None, normalize_input=True):
        """advanced_triplet...py:673-691: eval-mode segmentation (BatchNorm running statistics) of the min-max normalised input -> logits (or
        probabilities) [N,K,H,W].  Like the reference it leaves the solver in eval mode."""
        self.eval()
        with torch.no_grad():
            _, pred, _ = self.run(input, normalize_input=normalize_input)
        pred = pred.detach().clone()
        if softmax:
            pred = torch.softmax(pred, dim=1)
        return pred

    def update_schedule(self):
        for v in self.schedulers_dict.values():
            v.step()

    def set_running_metric(self):
        """advanced_triplet...py:1093-1095 (confusion matrix on the GPU: maxstyle_amd.metrics.runningScore)."""
        from .metrics import runningScore
        return runningScore(self.num_classes)       # the matrix is created on the device of the first update

    def evaluate(self, input, targets_npy, n_iter=None):
        """advanced_triplet...py:914-934: eval-mode prediction of one batch accumulated into self.running_metric; returns the logits."""
        if getattr(self, "running_metric", None) is None:
            self.running_metric = self.set_running_metric()
        self.flush_loop_errors()
        self.eval()
        pred = self.predict(input, n_iter=n_iter)
        targets = torch.as_tensor(targets_npy)
        self.running_metric.update(label_trues=targets, logits=pred)     # argmax fused into the confusion kernel (reference: pred.max(1)[1] on the host)
        self.cur_eval_images = input.detach().cpu().numpy()[:, 0, :, :]
        self.cur_eval_predicts = pred.max(1)[1].cpu().numpy()
        self.cur_eval_gts = targets.cpu().numpy()
        return pred

    # ------------------------------------------------------------------ the hot path
    def _weights_key(self):
        """Changes whenever a parameter or buffer of the three sub-nets was written through torch (in-place ops bump `_version`) or by the
        flat optimiser (`_weights_epoch`); cheap enough to evaluate on every call."""
        if self._wk_tensors is None:
            self._wk_tensors = [t for m in self.model.values() for t in list(m.parameters()) + list(m.buffers()) if t.is_floating_point()]
        return (self._weights_epoch, len(self._wk_tensors), sum(t._version for t in self._wk_tensors), id(self._wk_tensors[0]))

    _wk_tensors = None
    _bn_maps = None

    # storage type of the activation tensors of the INNER LOOP (generate_max_style_image): None / torch.float32, or torch.bfloat16 (BASELINE config 5's
    # "bf16 activations": conv inputs / outputs, gradients and the image are stored as bf16 inside the loop, statistics / parameters / arithmetic fp32; the
    # returned image is fp32 as in the reference).  Module forwards (predict / evaluate / training) are fp32.
    loop_act_dtype = None
    # True: this process shares its GPU with other processes / streams (several ranks per GPU): the inner loop must not select the single-read MaxStyle
    # kernel, whose grid assumes it gets every CU (engine.shared_device).  None: EngineOptions.shared_device (MS_SHARED_DEVICE; default off).
    loop_shared_device = None
    # "deferred" (the default since round 5; None): a spin time-out of the single-read kernel is reported by the NEXT call at the latest, and before anything lasting is
    # done with the image - optimize_all_params / optimize_params / evaluate / save_model / save_snapshots flush it (flush_loop_errors) - so the host prepares the next
    # call while the GPU still runs this one; "sync": MaxStyleHipError is raised by the very call that produced the invalid image (one event wait per call: the host
    # cannot run ahead, +0.8 ms per K = 5 call at config 2).
    loop_error_check = None
    loop_mfma_bf16 = None      # with bf16 storage: also bf16 matrix arithmetic in the 3x3 stride-1 convs (engine.mfma_bf16)
    # EngineOptions (maxstyle_amd/options.py; an object or a dict of its fields) of the engines this solver builds: the inner loop's / the training passes'.  Read when an
    # engine is built (one per shape): set them before the first call, or call `drop_engines()`.
    loop_options = None
    train_options = None

    def drop_engines(self):
        """Forget every engine (their buffers and captured graphs); the next call builds new ones with the current loop_options / train_options."""
        self._engines = {}
        self._train_engines = {}

    def _loop_engine(self, B, H, W, dev, act_dtype=torch.float32, mfma_bf16=False):
        key = self._weights_key()
        spec = E.NetSpec(self.reduce_factor, self.image_ch, self.num_classes)
        if self._packed_key != key:
            only_flat = (self._packed is not None and self._bank is not None and self._packed._desc is not None and self._packed_key is not None
                         and key[1:] == self._packed_key[1:])
            sd = None if only_flat else {k: {n: v.detach() for n, v in m.state_dict().items()} for k, m in self.model.items()}
            if self._packed is None:
                self._packed = E.PackedNets(spec, sd['image_encoder'], sd['segmentation_decoder'], sd['image_decoder'])
                for eng in self._engines.values():
                    eng.set_nets(self._packed)
                if self._bank is not None:
                    self._packed.bind_bank(self._bank)
            else:
                # the optimiser moved the weights: refresh the packed copies IN PLACE so captured HIP graphs stay valid
                if only_flat:
                    self._packed.repack_from_bank()          # one launch; biases / BatchNorm affine / heads alias the flat buffer
                else:
                    self._packed.update_(sd['image_encoder'], sd['segmentation_decoder'], sd['image_decoder'])
                for eng in self._engines.values():
                    eng._prefix_valid = False
            self._packed_key = key
        mfma_bf16 = bool(mfma_bf16) and act_dtype == torch.bfloat16
        ek = (B, H, W, str(dev)) if act_dtype == torch.float32 else (B, H, W, str(dev), str(act_dtype), mfma_bf16)
        eng = self._engines.get(ek)
        if eng is None:
            eng = E.InnerLoopEngine(spec, B, H, W, dev, act_dtype=act_dtype, mfma_bf16=mfma_bf16, options=self.loop_options)
            eng.set_nets(self._packed)
            self._engines[ek] = eng
        return eng

    def generate_max_style_image(self, image_code, decoder_layers_indexes=[3, 4, 5], channel_num=[128, 64, 32, 16, 16, 1], p=0.5,
                                 n_iter=5, mix_style=True, lr=0.1, no_noise=False, reference_image=None, reference_segmentation=None,
                                 noise_learnable=True, mix_learnable=True, loss_types=['seg'], loss_weights=[1], always_use_beta=False,
                                 debug=False, fix_seed=None, use_graph=True):
        """MaxStyle: K adversarial steps on {lmda, gamma_noise, beta_noise} of the layers inserted into the image decoder, returns the
        stylised reconstruction (detached clone). See the reference docstring (advanced_triplet...py:467-501) for the arguments."""
        if fix_seed is not None and isinstance(fix_seed, int):
            torch.manual_seed(fix_seed)
        if isinstance(image_code, list):
            raise NotImplementedError("list-valued codes belong to the (broken in the reference) Unet_im_recon path")
        if not len(decoder_layers_indexes) > 0:
            recon_image = self.decoder_inference(decoder_name='image_decoder', latent_code=image_code, disable_track_bn_stats=False)
            self.zero_grad()
            return recon_image.detach().clone()
        if not image_code.is_cuda:
            raise RuntimeError("generate_max_style_image runs on the MI355X only (HIP kernels); got a CPU tensor")
        old_state = {}
        for name, module in self.model.items():
            old_state[name] = module.training
            set_grad(module, requires_grad=False)
        try:
            batch_size = image_code.size(0)
            style_augmentor_dict = {}
            for i in decoder_layers_indexes:
                style_augmentor_dict[str(i)] = MaxStyle(batch_size, channel_num[i], p=p, mix_style=mix_style, no_noise=no_noise,
                                                        mix_learnable=mix_learnable, noise_learnable=noise_learnable,
                                                        always_use_beta=always_use_beta, debug=debug)
            nn_style_augmentor_dict = nn.ModuleDict(style_augmentor_dict)
            hook = getattr(self, "style_init_hook", None)       # test seam: inject perm / noise / lmda (SURVEY.md 7 'RNG')
            if hook is not None:
                hook(nn_style_augmentor_dict)
            optimize = True
            if n_iter > 0:
                if len(list(nn_style_augmentor_dict.parameters())) == 0:
                    optimize = False
                else:
                    assert reference_image is not None and reference_segmentation is not None, 'must provide reference images and segmentations'
                    self.zero_grad()
            for ltype in loss_types:
                if ltype != 'seg' and n_iter > 0 and optimize:
                    raise ValueError('loss type {} not supported'.format(ltype))
            code = image_code.detach().contiguous().float()
            B, _, h, w = code.shape
            act = self.loop_act_dtype or torch.float32
            mf = bool(self.loop_mfma_bf16)
            eng = self._loop_engine(B, h * 16, w * 16, code.device, act_dtype=act, mfma_bf16=mf)
            if self.loop_shared_device is not None:
                eng.shared_device = bool(self.loop_shared_device)
            mods = {int(k): m for k, m in nn_style_augmentor_dict.items()}
            slots = E.slots_from_modules(mods, code.device)
            layers = [i for i in sorted(mods) if i in slots]
            # everything a captured step bakes in as a launch argument or a kernel choice is part of the signature: the layer layout and flags,
            # eps, the Adam lr, the loss sign and BatchNorm batch-vs-running statistics (a replay with a different lr / loss weight / mode
            # silently reused the old arguments).  One captured graph is kept per signature (random-depth insertion alternates between a few).
            lr_ = float(lr)
            loss_sign = -float(sum(w_ for w_, t in zip(loss_weights, loss_types) if t == 'seg')) if optimize and n_iter > 0 else -1.0
            bn_eval = not all(old_state.values())
            sig = self._cfg_sig(layers, slots) + (lr_, loss_sign, bn_eval, bool(eng.shared_device))
            if not eng.restore_config(sig):
                eng.configure_styles(layers, slots)
            eng._cfg_sig = sig
            preset_std = False
            eng.set_style_states({i: (mods[i].perm, mods[i].lmda.detach(), mods[i].gamma_noise.detach(), mods[i].beta_noise.detach()) for i in layers})
            for i in layers:
                m = mods[i]
                if m.gamma_std is not None and m.beta_std is not None:       # (only a style_init_hook can have set them: maxstyle.py:165-168 keeps a std it already holds)
                    eng.preset_style_std(i, m.gamma_std, m.beta_std)
                    preset_std = True
            eng.lr = lr_
            eng.loss_sign = loss_sign
            eng.bn_eval = bn_eval
            labels = None
            if optimize and n_iter > 0:
                labels = reference_segmentation.to(device=code.device, dtype=torch.int64).contiguous()
            steps = n_iter if optimize else 0
            recon_image = eng.run(code, labels, steps, use_graph=use_graph and not preset_std)
            eng.stash_config(sig)
            mode = self.loop_error_check or "deferred"
            eng.check_errors(sync=(mode != "deferred"))                 # the single-read MaxStyle kernel's error word (spin time-out): never silent
            with torch.no_grad():                                  # hand the optimised state back to the modules (debugging / tests): two multi-tensor copies
                dsts, srcs = [], []
                for i in layers:
                    m = mods[i]
                    for nm in ("gamma_noise", "beta_noise", "lmda"):
                        d_, s_ = getattr(m, nm).data, eng.param(i, nm)
                        if d_.is_cuda and d_.dtype == s_.dtype and d_.shape == s_.shape:
                            dsts.append(d_); srcs.append(s_)
                        else:
                            d_.copy_(s_)
                    std = eng.buf.get(f"st{i}.std")
                    if std is not None:
                        m.gamma_std = torch.empty(1, std.shape[1], 1, 1, dtype=std.dtype, device=std.device); m.beta_std = torch.empty_like(m.gamma_std)
                        dsts += [m.gamma_std.view(-1), m.beta_std.view(-1)]; srcs += [std[0], std[1]]
                if dsts:
                    torch._foreach_copy_(dsts, srcs)
            self.last_style_modules = nn_style_augmentor_dict
            self.last_losses = eng.losses(steps).clone() if steps > 0 else None
            out = recon_image.detach().float().clone() if recon_image.dtype != torch.float32 else recon_image.detach().clone()
        finally:
            for name, module in self.model.items():
                set_grad(module, requires_grad=old_state[name])
        self.zero_grad()
        return out

    def generate_max_style_image_from_config(self, image_code, max_style_cfg, reference_image, reference_segmentation, p=0.5, rescale=False):
        """The trainer's call (train_adv...py:251-278) driven by the `max_style` block of a reference JSON config
        (config/ACDC/1500_epoch/MICCAI2022_MaxStyle.json:56-76), keys taken verbatim; `p=0.5` is the trainer's literal.
        rescale=True additionally applies rescale_intensity(.,0,1) as hard_example_traininng does next (advanced_triplet...py:868-869)."""
        if '16' in self.network_type:
            channel_num = [128, 64, 32, 16, 16, self.image_ch]
        elif '64' in self.network_type:
            channel_num = [512, 256, 128, 64, 64, self.image_ch]
        else:
            raise ValueError('network_type not supported')
        c = max_style_cfg
        out = self.generate_max_style_image(image_code=image_code, channel_num=channel_num, p=p,
                                            decoder_layers_indexes=c['decoder_layers_indexes'], n_iter=c['n_iter'], mix_style=c['mix_style'],
                                            lr=c['lr'], no_noise=c['no_noise'], reference_image=reference_image,
                                            reference_segmentation=reference_segmentation, noise_learnable=c['noise_learnable'],
                                            mix_learnable=c['mix_learnable'], loss_types=c['loss_types'], loss_weights=c['loss_weights'],
                                            always_use_beta=c.get('always_use_beta', False))
        if rescale:
            from . import ops
            out = ops.rescale_intensity(out.contiguous(), 0.0, 1.0)
        return out

    # ------------------------------------------------------------------ outer update (advanced_triplet...py:718-912, 1038-1091)
    _weights_epoch = 0

    def _param_bank(self):
        if self._bank is None:
            dev = next(self.model['image_encoder'].parameters()).device
            if dev.type != 'cuda':
                raise RuntimeError("training runs on the MI355X only (HIP kernels); move the solver to the GPU first")
            self._bank = T.ParamBank(self.model, dev)
            self._bank_sentinel = next(self.model[T.NETS[0]].parameters())       # first tensor of the flat buffers (offset 0)
            self._packed = None                       # rebuild the packed copies from tensors that alias the flat buffer
            self._packed_key = None
            self._anchor = torch.zeros(1, device=dev, requires_grad=True)
            self._wk_tensors = None
        else:
            p0 = self._bank_sentinel
            if p0.grad is None or p0.grad.data_ptr() != self._bank.flat_g.data_ptr():     # a zero_grad(set_to_none=True) dropped the views
                self._bank.rebind(self.model)
        return self._bank

    def set_optimizers(self):
        assert self.model
        bank = self._param_bank()
        self.optimizers = {name: _BankOptimizer(self, bank, name) for name in self.model}

    def reset_all_optimizers(self):
        if self.optimizers is None:
            self.set_optimizers()
        self._param_bank().zero_grad()

    def get_optimizer(self, model_name=None):
        assert self.optimizers, 'please set optimizers first before fetching'
        return self.optimizers if model_name is None else self.optimizers[model_name]

    def optimize_all_params(self):
        """advanced_triplet...py:1083-1085.  Under torch.distributed (one process per GPU) the flat gradient buffer is all-reduced once
        first - the data-parallel exchange the reference would get from DDP; set `self.data_parallel = False` to opt out."""
        self.flush_loop_errors()            # (deferred error protocol: never step the weights on a hard example that came from an invalid image)
        if getattr(self, "data_parallel", True) and self._bank is not None:
            self._bank.all_reduce_grads()
        for v in self.optimizers.values():
            v.step()

    def flush_loop_errors(self):
        """Resolve the deferred error check of every inner-loop engine (loop_error_check = "deferred", the default); a no-op in "sync" mode."""
        for eng in self._engines.values():
            eng.flush_errors()
        # the training engines' `_xfin` launches (cross-workgroup BatchNorm finalize in the forward passes) have an error word too: a weight step must not follow a
        # pass whose coefficients were zero-filled by a spin time-out - resolved here, in front of the step (the host waits for the passes it has queued)
        for pool in self._train_engines.values():
            for eng in pool:
                if "xfin.err" in eng.buf:
                    eng.check_errors(sync=True)

    def optimize_params(self, model_name):
        self.flush_loop_errors()          # (deferred error protocol: a hard example from a timed-out launch must not reach the weights - ADVICE r3)
        self.optimizers[model_name].step()

    def reset_optimizer(self, model_name):
        self.optimizers[model_name].zero_grad()

    # ------------------------------------------------------------------ checkpoints (same files and keys as the reference)
    def save_model(self, save_dir, epoch_iter, model_prefix=None, save_optimizers=False):
        """advanced_triplet...py:936-948: <save_dir>/<epoch_iter>/checkpoints/<net>.pth (+ <net>_optim.pth), plain state_dicts."""
        import os
        self.flush_loop_errors()
        epoch_path = os.path.join(save_dir, str(epoch_iter), 'checkpoints')
        os.makedirs(epoch_path, exist_ok=True)
        for model_name, model in self.model.items():
            torch.save(model.state_dict(), os.path.join(epoch_path, '{}.pth'.format(model_name)))
        if save_optimizers:
            if self.optimizers is None:
                self.set_optimizers()
            for model_name, optimizer in self.optimizers.items():
                torch.save(optimizer.state_dict(), os.path.join(epoch_path, '{}_optim.pth'.format(model_name)))

    def get_model_states_dict(self):
        """advanced_triplet...py:950-955."""
        return {model_name: model.state_dict() for model_name, model in self.model.items()}

    def restore_model(self, model_state_dict):
        """advanced_triplet...py:957-959 (in-place copies: the flat parameter buffers and captured graphs stay valid)."""
        for model_name, model_dict in model_state_dict.items():
            self.model[model_name].load_state_dict(model_dict)

    def save_snapshots(self, save_dir, epoch, model_prefix='interrupted'):
        """advanced_triplet...py:961-980: one .pkl with network_type / epoch / model_state / optimizer_state."""
        import os
        epoch_path = os.path.join(save_dir, 'interrupted', 'checkpoints')
        os.makedirs(epoch_path, exist_ok=True)
        model_prefix = self.network_type if model_prefix is None else self.network_type + model_prefix
        save_path = os.path.join(epoch_path, model_prefix + f'_{epoch}.pkl')
        if self.optimizers is None:
            self.set_optimizers()
        state = {'network_type': self.network_type, 'epoch': epoch, 'model_state': self.get_model_states_dict(),
                 'optimizer_state': {name: opt.state_dict() for name, opt in self.optimizers.items()}}
        torch.save(state, save_path)
        return save_path

    def load_snapshots(self, file_path):
        """advanced_triplet...py:982-1016: returns the epoch stored in the snapshot (0 when there is nothing to load)."""
        import os
        start_epoch = 0
        if file_path is None:
            return start_epoch
        if file_path == '' or not os.path.exists(file_path):
            print(f'warning: {file_path} does not exists')
            return start_epoch
        try:
            checkpoint = torch.load(file_path, map_location='cpu')
            if self.optimizers is None:
                self.set_optimizers()
            for k, v in self.model.items():
                v.load_state_dict(checkpoint['model_state'][k])
            for k, v in self.optimizers.items():
                v.load_state_dict(checkpoint['optimizer_state'][k])
            start_epoch = checkpoint['epoch']
            print("Loaded checkpoint '{}' (epoch {})".format(file_path, checkpoint['epoch']))
        except Exception as e:          # noqa: BLE001 - the reference reports and carries on
            print('error: {} in loading {}'.format(e, file_path))
        return start_epoch

    def compute_image_recon_loss(self, input_image, target, rec_loss_type=None):
        """0.5 * MSELoss(reduction='mean') of two device tensors (value only; the differentiable form is part of standard_training)."""
        from . import ops
        if (rec_loss_type or self.rec_loss_type) != 'l2':
            raise NotImplementedError
        return ops.mse_loss(input_image.contiguous().float(), target.contiguous().float())

    def _train_engine(self, B, H, W, dev):
        """A TrainEngine whose activations are not waiting for a backward (several passes can be alive before loss.backward())."""
        spec = E.NetSpec(self.reduce_factor, self.image_ch, self.num_classes)
        self._loop_engine(B, H, W, dev)                  # refreshes the packed weights if the optimiser moved them
        pool = self._train_engines.setdefault((B, H, W, str(dev)), [])
        for eng in pool:
            if not eng.pending:
                break
        else:
            eng = T.TrainEngine(spec, B, H, W, dev, options=self.train_options)
            eng.pending = False
            pool.append(eng)
        eng.nets = self._packed
        eng.bank = self._param_bank()
        return eng

    def fast_predict(self, input, domain_id=0, disable_track_bn_stats=False):
        """advanced_triplet...py:891-912 (value only: logits are detached; the differentiable route is standard_training)."""
        z_i, z_s = self.encode_image(input, domain_id, disable_track_bn_stats=disable_track_bn_stats)
        y_0 = self.decoder_inference(decoder=self.model['segmentation_decoder'], latent_code=z_s, disable_track_bn_stats=disable_track_bn_stats)
        return (z_i, z_s), y_0

    def standard_training(self, clean_image_l, label_l, perturbed_image, compute_gt_recon=True, update_latent=True, if_latent_code_consistency=False,
                          disable_track_bn_stats=False, domain_id=0, return_output=False, _enc_mix=None):
        """advanced_triplet...py:731-786 for 'no_STN' networks: seg loss = cross_entropy_2D(seg_decoder(z_s), labels), image recon loss =
        0.5*MSE(image_decoder(z_i), clean).  The two losses carry a grad_fn: `(a*seg + b*rec).backward()` runs the HIP backward pass and
        accumulates into the parameters' .grad.  recon_image / y_0 (return_output=True) and z_i / z_s are detached values."""
        if 'no_STN' not in self.network_type or 'no_im_recon' in self.network_type:
            raise NotImplementedError("shape-refinement (STN) and no_im_recon variants are outside the MaxStyle path (SURVEY.md 8)")
        if self.class_weights is not None:
            raise NotImplementedError("class_weights is None in every MaxStyle config")
        if not (self.training and all(m.training for m in self.model.values())):
            self.train()
        x = perturbed_image.detach().contiguous().float()
        if not x.is_cuda:
            raise RuntimeError("standard_training runs on the MI355X only (HIP kernels); got a CPU tensor")
        B, _, H, W = x.shape
        eng = self._train_engine(B, H, W, x.device)
        eng.enc_mix = _enc_mix                  # MixStyle / DSU layers inside the encoder (mixstyle_training); None for the plain passes
        labels = label_l.detach().to(device=x.device, dtype=torch.int64).contiguous()
        clean = clean_image_l.detach().contiguous().float()
        track = not disable_track_bn_stats
        if self._bn_maps is None:
            self._bn_maps = tuple({n: m for n, m in self.model[k].named_modules() if isinstance(m, nn.BatchNorm2d)} for k in T.NETS)
        bns = self._bn_maps
        if torch.is_grad_enabled():
            seg_loss, rec_loss = _TrainPassFn.apply(self._anchor, self, eng, x, labels, clean, track, bns)
        else:                                   # torch.no_grad(): values only - the engine's activations are not held for a backward
            eng.bn_affine_grad = track
            eng.run_forward(x, labels, clean, track, bns)
            vals = eng.loss_buf[:2].clone()
            seg_loss, rec_loss = vals[0], vals[1]
        z_i, z_s = eng._mixed(6, "e.z_i"), eng.buf["e.z_s"]
        if update_latent:
            self.z_i = z_i.clone()
            self.z_s = z_s.clone()
        self.recon_image = eng.buf["d.image"].clone()
        zero = torch.zeros((), device=x.device)                 # (torch.tensor(0., device=...) is a host-to-device copy: it waits for the stream)
        if return_output:
            y_0 = eng.buf["s.logits"].clone()
            return seg_loss, rec_loss, zero, zero, self.recon_image, y_0, y_0
        return seg_loss, rec_loss, zero, zero

    def hard_example_traininng(self, perturbed_image, clean_image_l, perturbed_seg, label_l, use_gpu=True, if_latent_code_consistency=False,
                               standard_input_image=None, standard_recon_image=None):
        """advanced_triplet...py:843-889 ('no_STN': only the image branch exists): rescale_intensity(perturbed, 0, 1), then
        standard_training inside _disable_tracking_bn_stats (batch statistics, no running update, BatchNorm affine frozen)."""
        from . import ops
        zero = torch.zeros((), device=clean_image_l.device)
        seg_loss, recon_loss, shape_loss = zero, zero, zero
        if perturbed_image is not None:
            if self.intensity_norm_type != 'min_max':
                raise NotImplementedError
            perturbed_image = ops.rescale_intensity(perturbed_image.detach().contiguous().float(), 0.0, 1.0)
            seg_loss, recon_loss, _, shape_loss = self.standard_training(clean_image_l=clean_image_l, label_l=label_l, perturbed_image=perturbed_image,
                                                                         compute_gt_recon=False, update_latent=False, disable_track_bn_stats=True, domain_id=0)
        return seg_loss, recon_loss, shape_loss, 0 * seg_loss

    # ------------------------------------------------------------------ MixStyle / DSU baselines inside the encoder
    def _draw_encoder_mixstyle(self, B, layers_indexes, lmda, mix, p, device, alpha=0.1, eps=1e-8):
        """The random draws of ONE MixStyle(p, alpha=0.1, lmda, mix) object called after inc (1), down1..down4 (2..5) and the final activation (6), in
        the reference's order (mixstyle.py:57-108: torch.rand(1) gate, Beta / constant lmda, then randperm or the two DSU normals on the device)."""
        r = self.reduce_factor
        chans = {1: 64 // r, 2: 128 // r, 3: 256 // r, 4: 512 // r, 5: 512 // r, 6: 512 // r}
        beta = torch.distributions.Beta(alpha, alpha)
        spec = {}
        for idx in range(1, 7):
            if idx not in layers_indexes:
                continue
            if torch.rand(1) > p:
                continue
            lm = beta.sample((B, 1, 1, 1)) if lmda is None else torch.ones(B, 1, 1, 1) * lmda
            lm = lm.to(device=device, dtype=torch.float32).contiguous()
            if mix in ('random', 'crossdomain'):
                if mix == 'random':
                    perm = torch.randperm(B)
                else:
                    perm = torch.arange(B - 1, -1, -1)
                    perm_b, perm_a = perm.chunk(2)
                    perm = torch.cat([perm_b[torch.randperm(B // 2)], perm_a[torch.randperm(B // 2)]], 0)
                spec[idx] = (perm.to(device=device, dtype=torch.int64).contiguous(), lm, None, None, eps)
            elif mix == 'gaussian':
                gmu = torch.randn(B, chans[idx], 1, 1, device=device)
                gstd = torch.randn(B, chans[idx], 1, 1, device=device)
                spec[idx] = (None, None, gstd.contiguous(), gmu.contiguous(), eps)
            else:
                raise NotImplementedError
        return spec

    def generate_style_augmented_latent_code(self, image, layers_indexes=[1, 2, 3], lmda=None, mix='random', p=0.5, _spec=None):
        """advanced_triplet...py:632-670: the encoder (batch statistics, running buffers untouched) with MixStyle / DSU after the chosen blocks ->
        (z_i, z_s) VALUES.  The differentiable form of the trainer's MixStyle branch is `mixstyle_training`."""
        x = image.detach().contiguous().float()
        if not x.is_cuda:
            raise RuntimeError("generate_style_augmented_latent_code runs on the MI355X only (HIP kernels); got a CPU tensor")
        B, _, H, W = x.shape
        eng = self._loop_engine(B, H, W, x.device)
        spec = self._draw_encoder_mixstyle(B, layers_indexes, lmda, mix, p, x.device) if _spec is None else _spec
        old_eval, eng.bn_eval = eng.bn_eval, False
        eng.enc_mix = spec
        try:
            with torch.no_grad():
                z_i, z_s = eng.encode_fwd(x)
                z_i, z_s = z_i.clone(), z_s.clone()
        finally:
            eng.enc_mix = None
            eng.bn_eval = old_eval
        self.latent_code['image'], self.latent_code['segmentation'] = z_i, z_s
        return z_i, z_s

    def mixstyle_training(self, clean_image_l, label_l, image, layers_indexes=[1, 2, 3], lmda=None, mix='random', p=0.5, _spec=None):
        """The trainer's MixStyle / DSU branch as ONE differentiable pass (train_adv...py:394-415): generate_style_augmented_latent_code ->
        segmentation decoder + image decoder inside _disable_tracking_bn_stats -> (cross_entropy_2D, 0.5*MSE to the clean image).  Both losses carry
        a grad_fn: backward runs through the decoders, the MixStyle layers (mu / sig detached) and the encoder."""
        x = image.detach().contiguous().float()
        spec = self._draw_encoder_mixstyle(x.shape[0], layers_indexes, lmda, mix, p, x.device) if _spec is None else _spec
        seg, rec, _, _ = self.standard_training(clean_image_l=clean_image_l, label_l=label_l, perturbed_image=x, compute_gt_recon=False, update_latent=False,
                                                disable_track_bn_stats=True, _enc_mix=spec)
        return seg, rec

    @staticmethod
    def _cfg_sig(layers, slots):
        return tuple((i, slots[i].B, slots[i].C, slots[i].mix_style, slots[i].use_noise, slots[i].learn_noise, slots[i].learn_mix, float(slots[i].eps)) for i in layers)


class _TrainPassFn(torch.autograd.Function):
    """Differentiable handle on one TrainEngine pass: forward returns the two scalar losses, backward runs the HIP backward pass (data and
    weight gradients) scaled by the upstream gradients and accumulates into the ParamBank - autograd itself never sees the networks."""

    @staticmethod
    def forward(ctx, anchor, solver, eng, x, labels, clean, track, bns):
        eng.bn_affine_grad = track          # _disable_tracking_bn_stats also freezes the BatchNorm affine (model_util.py:468-510)
        eng.run_forward(x, labels, clean, track, bns)
        eng.pending = True
        ctx.solver, ctx.eng = solver, eng
        # an output that takes no part in the caller's loss arrives in backward() as None, not as a materialised zero tensor: its branch of the backward pass is skipped
        # (a device-side 0 would still run it with scale 0, and 0 * inf of a diverged activation would put NaN into the ParamBank where the reference contributes
        # nothing - ADVICE r4).  A tensor-valued 0 the caller multiplies in itself still runs the branch.
        ctx.set_materialize_grads(False)
        out = eng.loss_buf[:2].clone()
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_seg, g_rec):
        eng, solver = ctx.eng, ctx.solver
        if not eng.pending:
            raise RuntimeError("the activations of this training pass were already consumed by a backward")
        # the upstream gradients stay on the device (TrainEngine.run_backward, the `_ds` seed launches): float(g) here would make the host wait for everything queued in
        # front of this backward - in a trainer iteration the whole inner loop - before it could issue the ~250 launches of the pass
        def dev_scalar(g):
            if g is None:
                return None
            return g.detach().to(dtype=torch.float32) if g.is_cuda else float(g)
        eng.bank = solver._param_bank()
        eng.run_backward(dev_scalar(g_seg), dev_scalar(g_rec))
        eng.pending = False
        return torch.zeros_like(solver._anchor), None, None, None, None, None, None, None


class _BankOptimizer:
    """Stands in for the reference's per-sub-net torch optimiser (advanced_triplet...py:1055-1086): AdamW / Adam on that net's slice of the flat buffers."""

    def __init__(self, solver, bank, net):
        self.solver, self.bank, self.net = solver, bank, net
        offs = [(o, n) for (k, _), (o, n, _) in bank.index.items() if k == net]
        self.begin = min(o for o, _ in offs)
        self.end = max(o + (n + 3) // 4 * 4 for o, n in offs)
        self.step_count = 0
        self.lr = solver.learning_rate

    def zero_grad(self, set_to_none=False):
        self.bank.flat_g[self.begin:self.end].zero_()

    def _params(self):
        """(offset, numel, shape) of this net's parameters in module.parameters() order - the index space of torch.optim state_dicts."""
        names = [n for n, _ in self.solver.model[self.net].named_parameters()]
        return [self.bank.index[(self.net, n)] for n in names]

    def state_dict(self):
        """Same layout as torch.optim.AdamW / Adam .state_dict() over this sub-net's parameters (what the reference stores in <net>_optim.pth)."""
        adamw = self.solver.optimizer_type == 'AdamW'
        state = {}
        if self.step_count > 0:
            for i, (o, n, shape) in enumerate(self._params()):
                state[i] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': self.bank.flat_m[o:o + n].view(shape).clone(),
                            'exp_avg_sq': self.bank.flat_v[o:o + n].view(shape).clone()}
        group = {'lr': self.lr, 'betas': (0.9, 0.999), 'eps': 1e-8, 'weight_decay': 1e-2 if adamw else 0.0, 'amsgrad': False, 'maximize': False,
                 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None, 'params': list(range(len(self._params())))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        params = self._params()
        groups = sd['param_groups']
        ids = [i for g in groups for i in g['params']]
        if len(ids) != len(params):
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
        self.lr = float(groups[0].get('lr', self.lr))
        steps = set()
        for pos, pid in enumerate(ids):
            st = sd['state'].get(pid)
            o, n, shape = params[pos]
            if st is None:
                self.bank.flat_m[o:o + n].zero_(); self.bank.flat_v[o:o + n].zero_()
                continue
            self.bank.flat_m[o:o + n].copy_(st['exp_avg'].reshape(-1)); self.bank.flat_v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise NotImplementedError("per-parameter step counts")
        self.step_count = steps.pop() if steps else 0

    def step(self):
        from ._lib import lib, check
        b = self.bank
        self.step_count += 1
        o, n = 4 * self.begin, self.end - self.begin
        wd = 1e-2 if self.solver.optimizer_type == 'AdamW' else 0.0
        check(lib.ms_adamw_step(b.flat_p.data_ptr() + o, b.flat_g.data_ptr() + o, b.flat_m.data_ptr() + o, b.flat_v.data_ptr() + o, n,
                                self.lr, 0.9, 0.999, 1e-8, wd, self.step_count, 0, torch.cuda.current_stream().cuda_stream), "ms_adamw_step")
        self.solver._weights_epoch += 1          # packed kernel-layout copies are refreshed in place at the next use
        self.solver.model[self.net].note_weights_changed()      # ... and the module path (encode_image / predict / evaluate) re-packs too
