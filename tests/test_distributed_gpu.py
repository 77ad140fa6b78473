"""The data-parallel outer step on real kernels: two processes share the one GPU of the test box (gloo moves the CUDA tensors; on a node the
same code runs one rank per GPU over RCCL - bench.py --gpus N).  Whole trainer iteration per rank on its own batch, ONE flat all-reduce inside
optimize_all_params(), rank-equal weights afterwards (train_adv_supervised_segmentation_triplet.py:532-535)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import maxstyle_amd as M
        from maxstyle_amd import distributed as D
        from maxstyle_amd import synthetic as syn
        torch.manual_seed(11 + rank)                             # different initial weights per rank on purpose
        S = M.AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard_no_STN", image_ch=1, num_classes=4, use_gpu=True,
                                                    optimizer_type="AdamW", learning_rate=1e-3)
        D.broadcast_parameters(list(S.model.values()), src=0)
        clean, lab = syn.synthetic_batch(4, 64, 1, 4, 1234 + rank)     # each rank its own shard
        clean, lab = clean.to(dev), lab.to(dev)
        cfg = {"mix_style": True, "no_noise": False, "lr": 0.1, "n_iter": 2, "mix_learnable": True, "noise_learnable": True,
               "decoder_layers_indexes": [3, 4, 5], "loss_types": ["seg"], "loss_weights": [1], "always_use_beta": False}
        S.train()
        S.reset_all_optimizers()
        seg0, rec0, gt0, sh0 = S.standard_training(clean, lab, perturbed_image=clean)
        S.reset_all_optimizers()
        sty = S.generate_max_style_image_from_config(S.z_i, cfg, clean, lab, p=1.5).detach().clone()
        seg1, rec1, sh1, sh2 = S.hard_example_traininng(perturbed_image=sty, perturbed_seg=None, clean_image_l=clean, label_l=lab)
        loss = (seg0 + rec0 + sh0 + gt0) + (rec1 + seg1 + sh1 + sh2)
        S.reset_all_optimizers()
        loss.backward()
        bank = S._param_bank()
        local = bank.flat_g.clone()
        both = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        w_before = bank.flat_p.clone()
        S.optimize_all_params()
        torch.cuda.synchronize()
        mean_g = sum(both) / world
        ok_grad = bool(torch.allclose(bank.flat_g, mean_g, rtol=0, atol=1e-6 * float(mean_g.abs().max())))
        differ = float((both[0] - both[1]).abs().max()) > 0      # the shards really produced different gradients
        w0 = bank.flat_p.clone()
        dist.broadcast(w0, 0)
        same_w = bool(torch.equal(bank.flat_p, w0))
        moved = float((bank.flat_p - w_before).abs().max())
        q.put((rank, ok_grad, differ, same_w, moved, float(loss.detach())))
    finally:
        dist.destroy_process_group()


def test_trainer_iteration_world2_on_one_gpu():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "flat gradient buffer must hold the mean over ranks after optimize_all_params()"
    assert all(r[2] for r in res)
    assert all(r[3] for r in res), "weights must be identical on every rank after the step"
    assert all(r[4] > 1e-5 for r in res)
    assert res[0][5] != res[1][5]
