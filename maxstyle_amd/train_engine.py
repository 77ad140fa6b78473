"""Outer training pass on one MI355X: forward with saved activations + full backward (data AND weight gradients) of
    encoder -> segmentation decoder -> cross entropy   and   encoder code z_i -> image decoder -> sigmoid -> 0.5*MSE
i.e. `standard_training` for 'no_STN' networks (/root/reference/src/models/advanced_triplet_recon_segmentation_model.py:731-786), which
`hard_example_traininng` (:843-889) re-runs on the stylised image inside `_disable_tracking_bn_stats`, followed by `loss.backward()` and the
optimiser steps of the trainer (train_adv_supervised_segmentation_triplet.py:532-535).  SURVEY.md 8(f) rows 1 and 3.

Same kernels and the same fusion plan as the inner loop (engine.py); what is added:
  * weight gradients: ms_conv_wgrad (pixel-reduction MFMA GEMM) reads the SAME tensors the data-gradient convs read - the masked gradient,
    the raw conv output (BatchNorm backward applied on the fly) and the producer's raw output (BatchNorm apply + LeakyReLU on the fly);
    nothing extra is materialised for them;
  * BatchNorm weight/bias gradients and the residual 1x1 conv's bias gradient fall out of the two per-channel sums the BatchNorm backward
    already needs (ms_bn_bwd_full);
  * biases of convolutions that feed a batch-statistics BatchNorm get an exact 0 gradient (the loss does not depend on them; the
    reference's value is summation round-off) - see oracle/outer_oracle.py:is_null_grad_bias;
  * all parameters / gradients / Adam moments of the three sub-nets live in ONE flat buffer each (ParamBank): zero_grad is one memset,
    the optimiser one launch, and the data-parallel all-reduce one collective over `flat_g` (distributed.FlatGradAllReduce).
Host code is orchestration only; there is no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import os

import torch

from . import ops
from ._lib import lib, check
from .engine import InnerLoopEngine, PackedNets, NetSpec, LEAKY, BN_EPS, F32

NETS = ("image_encoder", "segmentation_decoder", "image_decoder")
_TABLE = {"image_encoder": "enc", "segmentation_decoder": "seg", "image_decoder": "dec"}


class ParamBank:
    """Flat fp32 storage of every learnable tensor of the three sub-nets, in `NETS` x `named_parameters()` order (the order the reference's
    three optimisers would visit them).  `module.parameters()` become views of `flat_p`, their `.grad` views of `flat_g`."""

    def __init__(self, modules: Dict[str, torch.nn.Module], device):
        self.device = torch.device(device)
        self.index = {}
        total = 0
        plist = []
        for net in NETS:
            for name, p in modules[net].named_parameters():
                n = p.numel()
                self.index[(net, name)] = (total, n, tuple(p.shape))
                plist.append(p)
                total += (n + 3) // 4 * 4          # 16-byte aligned views
        self.total = total
        self.flat_p = torch.zeros(total, dtype=F32, device=self.device)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.step = 0
        with torch.no_grad():
            for (key, (off, n, shape)), p in zip(self.index.items(), plist):
                view = self.flat_p[off:off + n].view(shape)
                view.copy_(p.detach().to(self.device, F32))
                p.data = view
                p.grad = self.flat_g[off:off + n].view(shape)

    def grad(self, net, name) -> Optional[torch.Tensor]:
        e = self.index.get((net, name))
        if e is None:
            return None
        off, n, shape = e
        return self.flat_g[off:off + n].view(shape)

    def param(self, net, name):
        off, n, shape = self.index[(net, name)]
        return self.flat_p[off:off + n].view(shape)

    def zero_grad(self):
        self.flat_g.zero_()

    def rebind(self, modules):
        """Re-attach `.grad` views (a torch optimiser's zero_grad(set_to_none=True) drops them)."""
        for net in NETS:
            for name, p in modules[net].named_parameters():
                off, n, shape = self.index[(net, name)]
                if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                    p.grad = self.flat_g[off:off + n].view(shape)

    def all_reduce_grads(self, group=None, force=False):
        """Data-parallel exchange of the outer step: ONE all-reduce (sum, then x 1/world) over the flat gradient buffer, in
        place - the gradients already live in one contiguous buffer, so there is no pack / unpack.  RCCL ("nccl") on GPUs, gloo in the CPU tests."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1 and not force:        # force: issue the collective anyway (bench.py --force-dist / the RCCL self-test: the call path at world size 1)
            return
        dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=group)
        self.flat_g.mul_(1.0 / world)

    def optimizer_step(self, lr, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8, decoupled=True):
        """torch.optim.AdamW (decoupled) / torch.optim.Adam(weight_decay=0) on every parameter: one launch."""
        self.step += 1
        check(lib.ms_adamw_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(), self.total,
                                lr, betas[0], betas[1], eps, weight_decay if decoupled else 0.0, self.step, 0, torch.cuda.current_stream().cuda_stream),
              "ms_adamw_step")


class TrainEngine(InnerLoopEngine):
    """One forward/backward of the segmentation + reconstruction training pass.  Activations stay in this engine's buffers until
    `backward_pass`; use one TrainEngine per pass that is alive at the same time (standard pass, hard-example pass)."""

    def __init__(self, spec: NetSpec, B: int, H: int, W: int, device, options=None):
        super().__init__(spec, B, H, W, device, options=options)
        self.loss_sign = 1.0
        self.bank: Optional[ParamBank] = None
        self.bn_affine_grad = True       # False inside _disable_tracking_bn_stats (hard-example pass): BatchNorm weight/bias are constants
        self.bn_modules = None           # sd-name -> nn.BatchNorm2d per net, for the running-statistics update of a tracking pass
        self.track = False
        self._wg_calls = []              # (partial ptr, dst ptr, numel, slots) of the weight gradients issued in the current backward pass
        self._wg_desc = {}
        self._run_desc = {}
        # optional: from its third use on, a pass is ONE HIP-graph replay (first use allocates, second is captured).  Measured at C2: 29.5 vs 29.8 ms
        # per trainer iteration - the ~230 launches of a pass are not host-bound (4.5 ms of kernels in 6 ms of wall, the rest is launch boundaries
        # that a graph has too), so it is off by default (EngineOptions.train_graph turns it on).
        self.graph_passes = bool(self.options.train_graph)
        self._pgraphs = {}
        self.configure_styles([], {})

    # ------------------------------------------------------------------ helpers
    def _names(self, net):
        return self.nets._enc_names if net == "image_encoder" else PackedNets._dec_names(net == "image_decoder")

    def _gw(self, net, key):
        return self.bank.grad(net, self._names(net)[key] + ".weight")

    def _gb(self, net, key):
        return self.bank.grad(net, self._names(net)[key] + ".bias")

    def wgrad(self, p, q, dw, ks, stride=1, q_fetch=0, p_bnbwd=None, q_act=None):
        """Weight gradient of one conv: the MFMA kernel runs now (partials into this call's own region), the reduction into `dw` is deferred
        to ONE batched launch at the end of the backward pass (`_wgrad_flush`): ~70 reduce launches of 5 us each were 10 % of the pass."""
        N, M, Hp, Wp = p.shape
        Nq, Hq, Wq = q.shape[1:]
        nbytes = lib.ms_conv_wgrad_ws_bytes(N, M, Nq, Hp, Wp, ks, stride)
        idx = len(self._wg_calls)
        ws = self.t(f"wg.part{idx}", max(int(nbytes), 256), dtype=torch.uint8)
        pm, p2, pa, pb, pc = 0, 0, 0, 0, 0
        qm, qa, qb, slope = 0, 0, 0, 1.0
        if p_bnbwd is not None:
            pm = 2
            pa, pb, pc = ops.coef_ptrs(p_bnbwd[0])
            p2 = p_bnbwd[1].data_ptr()
        if q_act is not None:
            qm = 1
            qa, qb, _ = ops.coef_ptrs(q_act[0])
            slope = q_act[1]
        nslots = ctypes.c_int(0)
        check(lib.ms_conv_wgrad_partials(p.data_ptr(), p2, q.data_ptr(), N, M, Nq, Hp, Wp, Hq, Wq, ks, stride, q_fetch, pm, pa, pb, pc,
                                         qm, qa, qb, 4, slope, ws.data_ptr(), ws.numel(), ctypes.byref(nslots), self._st()), "ms_conv_wgrad_partials")
        self._wg_calls.append((ws.data_ptr(), dw.data_ptr(), dw.numel(), nslots.value))

    def _wgrad_flush(self):
        """One launch sums the partials of every weight tensor of this backward pass into the flat gradient buffer (accumulating)."""
        calls = self._wg_calls
        self._wg_calls = []
        if not calls:
            return
        key = tuple(calls)
        cached = self._wg_desc.get(key)
        if cached is None:
            import numpy as np
            assert lib.ms_wgrad_batch_desc_bytes() == 40
            dt = np.dtype([("block_begin", "<i8"), ("partial", "<u8"), ("dst", "<u8"), ("numel", "<i4"), ("nslots", "<i4"), ("accumulate", "<i4"), ("pad", "<i4")])
            rows, blocks = [], 0
            for part, dst, numel, nslots in calls:
                rows.append((blocks, part, dst, numel, nslots, 1, 0))
                blocks += (numel + 63) // 64
            table = torch.from_numpy(np.array(rows, dtype=dt).view(np.uint8).copy()).to(self.dev)
            cached = (table, len(rows), blocks)
            self._wg_desc = {key: cached}            # shapes are static: one table per engine (rebuilt only if the call sequence changes)
        table, n, blocks = cached
        check(lib.ms_wgrad_reduce_batch(table.data_ptr(), n, blocks, self._st()), "ms_wgrad_reduce_batch")

    def channel_sum(self, x, out):
        N, C, H, W = x.shape
        ws = self.t("cs.ws", max(lib.ms_channel_sum_ws_bytes(N, self.spec.code_ch), 64), dtype=torch.uint8)
        check(lib.ms_channel_sum(x.data_ptr(), N, C, H * W, out.data_ptr(), 1, ws.data_ptr(), ws.numel(), self._st()), "ms_channel_sum")

    def act_bwd_t(self, name, gin, ref, u, coef, slope, net, bn_key, dsum=None):
        """Mask gin in place, BatchNorm-backward coefficients, BatchNorm parameter gradients (+ dsum: bias.grad of the residual 1x1 conv)."""
        N, C, H, W = u.shape
        nparts = lib.ms_act_bwd_parts(N, C, H * W)
        part = self.t(name + ".part", C, nparts, 2)
        bc = self.t(name + ".bcoef", C, 4)
        check(lib.ms_act_bwd_reduce(gin.data_ptr(), 0 if ref is None else ref.data_ptr(), u.data_ptr(), coef.data_ptr(), gin.data_ptr(), part.data_ptr(),
                                    N, C, H * W, slope, self._st()), "ms_act_bwd_reduce:" + name)
        dg = db = None
        if self.bn_affine_grad:
            dg, db = self._gw(net, bn_key), self._gb(net, bn_key)
        check(lib.ms_bn_bwd_full(part.data_ptr(), nparts, coef.data_ptr(), float(N * H * W), bc.data_ptr(), 0 if dg is None else dg.data_ptr(),
                                 0 if db is None else db.data_ptr(), 0 if dsum is None else dsum.data_ptr(), 1, C, self._st()), "ms_bn_bwd_full:" + name)
        return gin, bc

    def dgrad_act_bwd_t(self, name, bw_name, g, cw, bnbwd, u, coef, slope, net, bn_key):
        """dgrad conv + act_bwd_t(ref=None) of the layer below; with `fuse_act_bwd` the mask and the sums come out of the conv's epilogue."""
        if not self.fuse_act_bwd:
            da, _, _ = self.conv(name, g, cw, bnbwd=bnbwd, dgrad=True)
            return self.act_bwd_t(bw_name, da, None, u, coef, slope, net, bn_key)
        out, tab = self.conv_actbwd(name, bw_name, g, cw, bnbwd, u, coef, slope)
        if not torch.is_tensor(tab):       # (a pending `_xfin` record: this engine reduces the table itself - ms_bn_bwd_full also yields the BatchNorm parameter gradients)
            tab = tab.tab
        N, C, H, W = u.shape
        bc = self.t(bw_name + ".bcoef", C, 4)
        dg = db = None
        if self.bn_affine_grad:
            dg, db = self._gw(net, bn_key), self._gb(net, bn_key)
        check(lib.ms_bn_bwd_full(tab.data_ptr(), 0, coef.data_ptr(), float(N * H * W), bc.data_ptr(), 0 if dg is None else dg.data_ptr(),
                                 0 if db is None else db.data_ptr(), 0, 1, C, self._st()), "ms_bn_bwd_full:" + bw_name)
        return out, bc

    def conv(self, name, x, cw, **kw):
        out, st, parts = super().conv(name, x, cw, **kw)
        if kw.get("stats"):
            self._last_count = out.shape[0] * out.shape[2] * out.shape[3]      # elements per channel the following BatchNorm sees
        return out, st, parts

    def conv_ups2(self, name, x, cw, fin=None):
        out, st, parts = super().conv_ups2(name, x, cw, fin=fin)
        self._last_count = out.shape[0] * out.shape[2] * out.shape[3]
        return out, st, parts

    def bn_fin(self, name, st, parts, bn):
        coef = super().bn_fin(name, st, parts, bn)
        self._coef_made(bn, coef)
        return coef

    def _coef_made(self, bn, coef):
        # (every BatchNorm of a tracking pass exactly once, in layer order: the statistics conv in front set _last_count)
        if self.track and not self.bn_eval:
            self._tracked.append((bn.name, coef, self._last_count))

    # ------------------------------------------------------------------ graph-replayed passes
    def _static_inputs(self, image, labels, clean):
        xi = self.t("in.image", *image.shape); xi.copy_(image)
        li = self.t("in.labels", *labels.shape, dtype=torch.int64); li.copy_(labels)
        ci = self.t("in.clean", *clean.shape); ci.copy_(clean)
        return xi, li, ci

    def _replayed(self, key, fn):
        """fn() issues only pre-allocated-buffer kernel launches on the current stream: eager the first time (allocations, descriptor tables),
        captured the second time, replayed afterwards.  Returns fn's result (engine buffers: the same objects every time)."""
        st = self._pgraphs.get(key)
        if st is None:
            out = fn()
            self._pgraphs[key] = "warm"
            return out
        if st == "warm":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            st = (g, out)
            self._pgraphs[key] = st
        st[0].replay()
        return st[1]

    def run_forward(self, image, labels, clean, track: bool, net_bns=None):
        """forward_pass on engine-owned copies of the inputs (a captured graph holds addresses); see `graph_passes`."""
        xi, li, ci = self._static_inputs(image, labels, clean)
        if not self.graph_passes or self.enc_mix is not None:        # (the MixStyle layers allocate their outputs: not capturable)
            return self.forward_pass(xi, li, ci, track, net_bns)
        key = ("fwd", bool(track), bool(self.bn_affine_grad), id(self.nets), None if net_bns is None else tuple(id(d) for d in net_bns))
        out = self._replayed(key, lambda: self.forward_pass(xi, li, ci, track, net_bns))
        if track and net_bns is not None:
            self.nets.eval_dirty = True      # host-side effect of forward_pass that a replay does not repeat
        self.track = False
        return out

    def run_backward(self, g_seg, g_rec):
        """backward_pass of the pass last run by run_forward (same static inputs).  g_seg / g_rec: the upstream gradients of the two losses - host floats (baked into
        the launches) or 0-dim fp32 CUDA tensors (read by the seed kernels on the device: no host wait for the queue in front of the backward); None / 0.0: branch skipped."""
        b = self.buf
        xi, li, ci = b["in.image"], b["in.labels"], b["in.clean"]
        if torch.is_tensor(g_seg) or torch.is_tensor(g_rec):
            # engine-owned copies: the launches (and a captured graph) keep these addresses
            def own(name, g):
                if g is None:
                    return None
                t = self.t(name, 1)
                t.copy_(g.detach().reshape(1) if torch.is_tensor(g) else torch.full((1,), float(g), device=self.dev))
                return t
            g_seg, g_rec = own("in.g_seg", g_seg), own("in.g_rec", g_rec)
            if not self.graph_passes or self.enc_mix is not None:
                return self.backward_pass(xi, li, ci, g_seg, g_rec)
            key = ("bwd", g_seg is not None, g_rec is not None, "dev", bool(self.bn_affine_grad), id(self.nets), id(self.bank))
            return self._replayed(key, lambda: self.backward_pass(xi, li, ci, g_seg, g_rec))
        if not self.graph_passes or self.enc_mix is not None:
            return self.backward_pass(xi, li, ci, g_seg, g_rec)
        key = ("bwd", None if g_seg is None else float(g_seg), None if g_rec is None else float(g_rec), bool(self.bn_affine_grad), id(self.nets), id(self.bank))
        self._replayed(key, lambda: self.backward_pass(xi, li, ci, g_seg, g_rec))

    # ------------------------------------------------------------------ forward
    def forward_pass(self, image, labels, clean, track: bool, net_bns=None):
        """Returns (z_i, logits, recon); losses in loss_buf[0] (cross entropy) and loss_buf[1] (0.5*MSE)."""
        self._prefix_valid = False
        self.track = track
        self._tracked = []
        self.bn_eval = False
        self.bn_observer = None
        e_bns, s_bns, d_bns = net_bns if net_bns is not None else (None, None, None)
        z_i, z_s = self.encode_fwd(image)
        n_enc = len(self._tracked)
        h = self.seg_fwd(z_s)
        n_seg = len(self._tracked)
        N, C, H, W = h.shape
        w, bias = self.nets.seg["head.w"], self.nets.seg["head.b"]
        K = w.shape[0]
        logits = self.t("s.logits", N, K, H, W)
        ws = self.t("s.ce_ws", max(lib.ms_head_ce_ws_bytes(N, H * W), 64), dtype=torch.uint8)
        check(lib.ms_head_ce(h.data_ptr(), w.data_ptr(), bias.data_ptr(), labels.data_ptr(), 0, logits.data_ptr(), self.loss_buf.data_ptr(), 0,
                             N, C, K, H * W, 1.0, ws.data_ptr(), ws.numel(), self._st()), "ms_head_ce")
        self.buf["s.head_in"] = h
        recon = self.decode(z_i)
        mws = self.t("d.mse_ws", lib.ms_mse_ws_bytes(), dtype=torch.uint8)
        n = recon.numel()
        check(lib.ms_mse_loss(recon.data_ptr(), clean.data_ptr(), n, 0.5 / n, 0.0, self.loss_buf.data_ptr() + 4, 0, mws.data_ptr(), mws.numel(), self._st()), "ms_mse_loss")
        if track and net_bns is not None:
            self._update_running(self._tracked[:n_enc], e_bns)
            self._update_running(self._tracked[n_enc:n_seg], s_bns)
            self._update_running(self._tracked[n_seg:], d_bns)
        self.track = False
        return z_i, logits, recon

    def _update_running(self, tracked, bns):
        """Running statistics of every BatchNorm of one sub-net: one launch (momentum 0.1 everywhere in these networks) + one foreach add."""
        if not tracked:
            return
        key = tuple((name, coef.data_ptr()) for name, coef, _ in tracked)
        cached = self._run_desc.get(key)
        if cached is None:
            import numpy as np
            assert lib.ms_bn_running_desc_bytes() == 32
            dt = np.dtype([("coef", "<u8"), ("rm", "<u8"), ("rv", "<u8"), ("C", "<i4"), ("count", "<f4")])
            rows, counters, mom = [], [], None
            for name, coef, u_count in tracked:
                m = bns[name]
                mm = 0.1 if m.momentum is None else float(m.momentum)
                if mom is not None and mm != mom:
                    raise NotImplementedError("per-layer BatchNorm momentum")
                mom = mm
                rows.append((coef.data_ptr(), m.running_mean.data_ptr(), m.running_var.data_ptr(), coef.shape[0], float(u_count)))
                counters.append(m.num_batches_tracked)
            table = torch.from_numpy(np.array(rows, dtype=dt).view(np.uint8).copy()).to(self.dev)
            cached = (table, len(rows), mom, counters)
            self._run_desc[key] = cached
        table, n, mom, counters = cached
        check(lib.ms_bn_running_update_batch(table.data_ptr(), n, mom, BN_EPS, self._st()), "ms_bn_running_update_batch")
        torch._foreach_add_(counters, 1)
        self.nets.eval_dirty = True          # eval-mode BatchNorm tables are recomputed from the running statistics on their next use

    # ------------------------------------------------------------------ backward
    def res_bwd_t(self, pfx, net, key, x, dout, kind, need_dx=True):
        """Backward of one residual block with weight gradients.  x: the block's input tensor; dout: gradient w.r.t. its output (overwritten)."""
        b = self.buf
        tbl = getattr(self.nets, _TABLE[net])
        c0, c3, ci = tbl[key + ".c0"], tbl[key + ".c3"], tbl[key + ".ci"]
        u1, u2 = b[pfx + ".u1"], b[pfx + ".u2"]
        cf1 = b[pfx + ".bn1.coef"]
        g2, bc2 = self.act_bwd_t(pfx + ".bw2", dout, b[pfx + ".out"], u2, b[pfx + ".bn4.coef"], LEAKY, net, key + ".bn4", dsum=self._gb(net, key + ".ci"))
        self.wgrad(g2, u1, self._gw(net, key + ".c3"), 3, p_bnbwd=(bc2, u2), q_act=(cf1, LEAKY))
        g1, bc1 = self.dgrad_act_bwd_t(pfx + ".da1", pfx + ".bw1", g2, c3, (bc2, u2), u1, cf1, LEAKY, net, key + ".bn1")
        src = x if kind == "nn" else (b[pfx + ".xu"] if kind == "convT" else b[pfx + ".xd"])
        self.wgrad(g1, src, self._gw(net, key + ".c0"), 3, q_fetch=1 if kind == "nn" else 0, p_bnbwd=(bc1, u1))
        dsrc, _, _ = self.conv(pfx + ".dsrc", g1, c0, bnbwd=(bc1, u1), dgrad=True)
        if kind == "nn":
            dx = self.pool2(pfx + ".dx", dsrc)
            gs = self.pool2(pfx + ".gs", g2)
            self.wgrad(gs, x, self._gw(net, key + ".ci"), 1)
            self.conv(pfx + ".dx", gs, ci, dgrad=True, epi=1, out=dx)
            return dx
        self.wgrad(g2, src, self._gw(net, key + ".ci"), 1)
        self.conv(pfx + ".dsrc", g2, ci, dgrad=True, epi=1, out=dsrc)
        if kind == "convT":
            self.wgrad(x, dsrc, self._gw(net, key + ".up"), 2, stride=2)
            self.channel_sum(dsrc, self._gb(net, key + ".up"))
            if not need_dx:
                return None
            dx, _, _ = self.conv(pfx + ".dx", dsrc, tbl[key + ".up"], ks=2, stride=2, dgrad=True)
        else:
            self.wgrad(dsrc, x, self._gw(net, key + ".down"), 3, stride=2)
            self.channel_sum(dsrc, self._gb(net, key + ".down"))
            if not need_dx:
                return None
            dx = self.dgrad_s2(pfx + ".dx", dsrc, tbl[key + ".down"])
        return dx

    def backward_pass(self, image, labels, clean, g_seg, g_rec):
        """Accumulates d(g_seg*CE + g_rec*0.5*MSE)/d(parameters) into the ParamBank's flat gradient buffer (g_*: host floats or 1-element CUDA tensors, see run_backward)."""
        b = self.buf
        ds_seg = g_seg.data_ptr() if torch.is_tensor(g_seg) else 0        # device-side upstream gradients (the `_ds` seed launches)
        ds_rec = g_rec.data_ptr() if torch.is_tensor(g_rec) else 0
        if torch.is_tensor(g_seg):
            g_seg = 1.0
        if torch.is_tensor(g_rec):
            g_rec = 1.0
        g_seg = 0.0 if g_seg is None else g_seg
        g_rec = 0.0 if g_rec is None else g_rec
        self._wg_calls = []
        net_e, net_s, net_d = NETS
        e, s, d = self.nets.enc, self.nets.seg, self.nets.dec
        z_i, z_s = self._mixed(6, "e.z_i"), b["e.z_s"]          # z_i: what the image decoder and the code_decoupler consumed
        # ---- segmentation branch: cross entropy -> head -> four 'NN' up blocks -> dz_s
        dz_s = None
        if g_seg != 0.0:
            h = b["s.head_in"]
            N, C, H, W = h.shape
            w, bias = s["head.w"], s["head.b"]
            K = w.shape[0]
            dh = self.t("s.dh", N, C, H, W)
            ws = b["s.ce_ws"]
            scratch = self.t("s.loss_scratch", 4)
            check(lib.ms_head_ce_ds(h.data_ptr(), w.data_ptr(), bias.data_ptr(), labels.data_ptr(), dh.data_ptr(), 0, scratch.data_ptr(), 0,
                                    N, C, K, H * W, float(g_seg), ds_seg, ws.data_ptr(), ws.numel(), self._st()), "ms_head_ce(bwd)")
            hws = self.t("s.hw_ws", max(lib.ms_head_wgrad_ws_bytes(N, C, K, H * W), 64), dtype=torch.uint8)
            check(lib.ms_head_wgrad_ds(h.data_ptr(), b["s.logits"].data_ptr(), labels.data_ptr(), 0, float(g_seg) / (N * H * W), ds_seg,
                                       self.bank.grad(net_s, "final_conv.weight").data_ptr(), self.bank.grad(net_s, "final_conv.bias").data_ptr(),
                                       N, C, K, H * W, 1, hws.data_ptr(), hws.numel(), self._st()), "ms_head_wgrad(seg)")
            g = dh
            for i in range(4, 0, -1):
                x = z_s if i == 1 else b[f"s.u{i - 1}.out"]
                g = self.res_bwd_t(f"s.u{i}", net_s, f"u{i}", x, g, "nn")
            dz_s = g
        # ---- reconstruction branch: 0.5*MSE -> sigmoid head -> four ConvTranspose up blocks -> dz_i
        dz_i = None
        if g_rec != 0.0:
            img = b["d.image"]
            x = b["d.head_in"]
            N, C, H, W = x.shape
            K = d["head.w"].shape[0]
            n = img.numel()
            hws = self.t("d.hw_ws", max(lib.ms_head_wgrad_ws_bytes(N, C, K, H * W), 64), dtype=torch.uint8)
            check(lib.ms_head_wgrad_ds(x.data_ptr(), img.data_ptr(), clean.data_ptr(), 1, float(g_rec) / n, ds_rec,
                                       self.bank.grad(net_d, "final_conv.weight").data_ptr(), self.bank.grad(net_d, "final_conv.bias").data_ptr(),
                                       N, C, K, H * W, 1, hws.data_ptr(), hws.numel(), self._st()), "ms_head_wgrad(dec)")
            dimg = self.t("d.dimg", *img.shape)
            mws = b["d.mse_ws"]
            check(lib.ms_mse_loss_ds(img.data_ptr(), clean.data_ptr(), n, 0.0, float(g_rec) / n, ds_rec, 0, dimg.data_ptr(), mws.data_ptr(), mws.numel(), self._st()), "ms_mse_loss(bwd)")
            dh = self.t("d.dh", N, C, H, W)
            check(lib.ms_head_bwd(dimg.data_ptr(), img.data_ptr(), d["head.w"].data_ptr(), dh.data_ptr(), N, C, K, H * W, 1, self._st()), "ms_head_bwd")
            g = dh
            for i in range(4, 0, -1):
                x = z_i if i == 1 else b[f"d.u{i - 1}.out"]
                g = self.res_bwd_t(f"d.u{i}", net_d, f"u{i}", x, g, self._dec_kind())
            dz_i = g
        # ---- encoder: code_decoupler (z_s branch) joins the image-decoder gradient at z_i
        if dz_s is not None:
            g, bc = self.act_bwd_t("e.cd.bw2", dz_s, z_s, b["e.cd.u2"], b["e.cd.bn4.coef"], 0.0, net_e, "cd4")
            self.wgrad(g, b["e.cd.u1"], self._gw(net_e, "cd3"), 3, p_bnbwd=(bc, b["e.cd.u2"]), q_act=(b["e.cd.bn1.coef"], LEAKY))
            g, bc = self.dgrad_act_bwd_t("e.cd.da", "e.cd.bw1", g, e["cd3"], (bc, b["e.cd.u2"]), b["e.cd.u1"], b["e.cd.bn1.coef"], LEAKY, net_e, "cd1")
            self.wgrad(g, z_i, self._gw(net_e, "cd0"), 3, p_bnbwd=(bc, b["e.cd.u1"]))
            if dz_i is None:
                dz_i, _, _ = self.conv("e.dz_i", g, e["cd0"], bnbwd=(bc, b["e.cd.u1"]), dgrad=True)
            else:
                self.conv("e.dz_i", g, e["cd0"], bnbwd=(bc, b["e.cd.u1"]), dgrad=True, epi=1, out=dz_i)
        if dz_i is None:
            self._wgrad_flush()
            return
        hin = self._mixed(5, "e.d4.out")
        dz_i = self._mix_bwd(6, dz_i, b["e.z_i"])
        g, bc = self.act_bwd_t("e.fc.bw", dz_i, b["e.z_i"], b["e.fc.u"], b["e.fc.bn.coef"], 0.0, net_e, "fc1")
        self.wgrad(g, hin, self._gw(net_e, "fc0"), 1, p_bnbwd=(bc, b["e.fc.u"]))
        dh, _, _ = self.conv("e.fc.dh", g, e["fc0"], bnbwd=(bc, b["e.fc.u"]), dgrad=True)
        dh = self._mix_bwd(5, dh, b["e.d4.out"])
        for i in range(4, 0, -1):
            prev = "e.inc.out" if i == 1 else f"e.d{i - 1}.out"
            dh = self.res_bwd_t(f"e.d{i}", net_e, f"d{i}", self._mixed(i, prev), dh, "down")
            dh = self._mix_bwd(i, dh, b[prev])
        g, bc = self.act_bwd_t("e.inc.bw2", dh, b["e.inc.out"], b["e.inc.ub"], b["e.inc.bn4.coef"], LEAKY, net_e, "inc4")
        self.wgrad(g, b["e.inc.ua"], self._gw(net_e, "inc3"), 3, p_bnbwd=(bc, b["e.inc.ub"]), q_act=(b["e.inc.bn1.coef"], LEAKY))
        g, bc = self.dgrad_act_bwd_t("e.inc.da", "e.inc.bw1", g, e["inc3"], (bc, b["e.inc.ub"]), b["e.inc.ua"], b["e.inc.bn1.coef"], LEAKY, net_e, "inc1")
        self.wgrad(g, image, self._gw(net_e, "inc0"), 3, p_bnbwd=(bc, b["e.inc.ua"]))      # the input image needs no gradient
        self._wgrad_flush()
