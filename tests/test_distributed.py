"""N>1 path on CPU: gloo, world_size 2 (the GPU box runs the same code over RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import maxstyle_amd as M
        from maxstyle_amd import distributed as D
        torch.manual_seed(100 + rank)                       # different initial weights per rank on purpose
        S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", use_gpu=False)
        mods = list(S.model.values())
        D.broadcast_parameters(mods, src=0)
        w0 = torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()])
        ref = w0.clone(); dist.broadcast(ref, 0)
        same = bool(torch.equal(w0, ref))
        # rank-dependent gradients -> averaged by ONE flat all-reduce
        for m in mods:
            for i, p in enumerate(m.parameters()):
                p.grad = torch.full_like(p, float(rank + 1) * (1 + (i % 3)))
        bucket = D.FlatGradAllReduce(mods)
        assert bucket.numel == 1536325                      # FCN_16: 1 125 632 + 161 732 + 248 961
        bucket.reduce()
        ok = True
        for m in mods:
            for i, p in enumerate(m.parameters()):
                ok = ok and bool(torch.allclose(p.grad, torch.full_like(p, 1.5 * (1 + (i % 3)))))
        # the outer update's own exchange: the ParamBank holds every gradient in one flat buffer -> one in-place all-reduce, no packing
        from maxstyle_amd.train_engine import ParamBank
        bank = ParamBank(S.model, "cpu")
        assert bank.total >= 1536325 and all(p.grad.data_ptr() >= bank.flat_g.data_ptr() for m in mods for p in m.parameters())
        for m in mods:
            for i, p in enumerate(m.parameters()):
                p.grad.fill_(float(rank + 1) * (1 + (i % 5)))
        bank.all_reduce_grads()
        ok = ok and all(bool(torch.allclose(p.grad, torch.full_like(p, 1.5 * (1 + (i % 5))))) for m in mods for i, p in enumerate(m.parameters()))
        ok = ok and bool(torch.equal(torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()]), w0))      # parameters became views, values kept
        lo, hi = D.shard_range(32, rank, world)
        tmax = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
        q.put((rank, same, ok, (lo, hi), tmax))
    finally:
        dist.destroy_process_group()


def test_flat_allreduce_and_broadcast_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] and r[2] for r in res)
    assert res[0][3] == (0, 16) and res[1][3] == (16, 32)
    assert all(abs(r[4] - 2.0) < 1e-12 for r in res)


def _torch_adamw(self):
    """TEST-ONLY stand-in for the ms_adamw_step HIP kernel so that optimize_all_params() can run end to end on the CPU box (the product has
    no CPU path: _BankOptimizer.step launches the kernel).  torch.optim.AdamW arithmetic on this sub-net's slice of the flat buffers."""
    b = self.bank
    self.step_count += 1
    sl = slice(self.begin, self.end)
    g, m, v, p = b.flat_g[sl], b.flat_m[sl], b.flat_v[sl], b.flat_p[sl]
    wd = 1e-2 if self.solver.optimizer_type == 'AdamW' else 0.0
    p.mul_(1 - self.lr * wd)
    m.mul_(0.9).add_(g, alpha=0.1)
    v.mul_(0.999).addcmul_(g, g, value=0.001)
    bc1, bc2 = 1 - 0.9 ** self.step_count, 1 - 0.999 ** self.step_count
    p.addcdiv_(m, (v.sqrt() / bc2 ** 0.5).add_(1e-8), value=-self.lr / bc1)
    self.solver._weights_epoch += 1


def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import maxstyle_amd as M
        from maxstyle_amd import distributed as D
        from maxstyle_amd import solver as SV
        from maxstyle_amd.train_engine import ParamBank
        torch.manual_seed(7 + rank)
        S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", use_gpu=False, optimizer_type="AdamW", learning_rate=1e-3)
        D.broadcast_parameters(list(S.model.values()), src=0)
        S._bank = ParamBank(S.model, "cpu")
        S.optimizers = {name: SV._BankOptimizer(S, S._bank, name) for name in S.model}
        SV._BankOptimizer.step = _torch_adamw
        w_before = S._bank.flat_p.clone()
        gen = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(S._bank.total, generator=gen)
        S._bank.flat_g.copy_(local)
        both = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        S.optimize_all_params()                              # flat all-reduce (mean) -> three optimiser steps
        mean_g = sum(both) / world
        ok_grad = bool(torch.allclose(S._bank.flat_g, mean_g, rtol=0, atol=1e-7))
        w0 = S._bank.flat_p.clone()
        dist.broadcast(w0, 0)
        same_w = bool(torch.equal(S._bank.flat_p, w0))      # every rank made the same step from the same weights
        moved = float((S._bank.flat_p - w_before).abs().max())
        views = all(p.data_ptr() >= S._bank.flat_p.data_ptr() for m in S.model.values() for p in m.parameters())
        q.put((rank, ok_grad, same_w, moved, views))
    finally:
        dist.destroy_process_group()


def test_optimize_all_params_world2_keeps_ranks_in_step():
    """solver.optimize_all_params() under torch.distributed (train_adv_supervised_segmentation_triplet.py:532-535 with the DDP exchange the reference
    gets from its launcher): rank-dependent gradients -> ONE flat all-reduce -> identical weights on both ranks after the step."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "flat gradient buffer must hold the mean over ranks"
    assert all(r[2] for r in res), "weights must be identical on every rank after the step"
    assert all(r[3] > 1e-5 for r in res) and all(r[4] for r in res)
