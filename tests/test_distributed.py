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
