"""Data-parallel plumbing for the surrounding training step.

The inner MaxStyle loop has NO collective: every rank optimises the style parameters of its own batch (perm, gamma_std/beta_std and
BatchNorm batch statistics are per-batch quantities), so N GPUs = N independent replicas of the engine.  The one exchange of the
outer step is the gradient of the three sub-nets (`image_encoder`, `segmentation_decoder`, `image_decoder`; 1.54 M fp32 = 6.1 MB for
FCN_16, ~98 MB for FCN_64) between `loss.backward()` and the optimiser steps (train_adv...py:534-535): ONE flat all-reduce
(sum, then x 1/world) instead of one collective per tensor - on MI355X xGMI is point-to-point (7 links x ~153 GB/s per GPU), the
6 MB buffer is latency-bound, so fewer, larger messages are what matters.  backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def _visible_list(value: str) -> int:
    """Number of devices a *_VISIBLE_DEVICES value selects: comma-separated indexes / UUIDs, cut at the first empty or negative entry
    (the runtime's rule: everything after an invalid entry is ignored)."""
    n = 0
    for tok in value.split(","):
        tok = tok.strip()
        if not tok or tok.startswith("-"):
            break
        n += 1
    return n


def visible_gpu_count(kfd_nodes: str = KFD_NODES, env: Optional[Dict[str, str]] = None, dri: str = "/dev/dri") -> int:
    """GPUs this process tree may use, WITHOUT touching HIP / torch.cuda: KFD topology nodes that have SIMDs (CPU nodes have simd_count 0)
    and whose render node this process can open (a container sees every GPU of the host in sysfs but only its own /dev/dri/renderD*),
    narrowed by ROCR_VISIBLE_DEVICES and then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  For the parent of `bench.py --gpus N`, which must
    stay free of any GPU call before it starts its ranks (a GPU-initialised process is never forked or exec'ed).  Returns -1 when the
    topology is not readable (not a ROCm box): the caller then skips its check rather than guess."""
    env = os.environ if env is None else env
    try:
        nodes = sorted(os.listdir(kfd_nodes), key=lambda s: (len(s), s))
    except OSError:
        return -1
    n = 0
    for node in nodes:
        try:
            props = {}
            with open(os.path.join(kfd_nodes, node, "properties")) as f:
                for line in f:
                    k, _, v = line.partition(" ")
                    props[k] = v.strip()
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
            if minor >= 0 and os.path.isdir(dri) and not os.access(os.path.join(dri, f"renderD{minor}"), os.R_OK | os.W_OK):
                continue
            n += 1
        except (OSError, ValueError):
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        if var in env:
            n = min(n, _visible_list(env[var]))
    if "HIP_VISIBLE_DEVICES" not in env and "CUDA_VISIBLE_DEVICES" in env:      # HIP honours the CUDA spelling when its own is unset
        n = min(n, _visible_list(env["CUDA_VISIBLE_DEVICES"]))
    return n


def _params(modules: Iterable[torch.nn.Module]) -> List[torch.nn.Parameter]:
    return [p for m in modules for p in m.parameters()]


def broadcast_parameters(modules: Iterable[torch.nn.Module], src: int = 0) -> None:
    """Start every rank from rank `src`'s weights and buffers (the reference has no SyncBN: running statistics then evolve per rank)."""
    tensors = [p.data for p in _params(modules)] + [b for m in modules for b in m.buffers()]
    if not tensors:
        return
    floats = [t for t in tensors if t.is_floating_point()]
    flat = torch.cat([t.reshape(-1) for t in floats])
    dist.broadcast(flat, src)
    off = 0
    for t in floats:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    for t in tensors:
        if not t.is_floating_point():
            dist.broadcast(t, src)


class FlatGradAllReduce:
    """One persistent flat fp32 bucket for the outer gradients; `reduce()` = pack, ONE all-reduce, average, unpack."""

    def __init__(self, modules: Iterable[torch.nn.Module]):
        self.params = [p for p in _params(modules) if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.numel = n

    def reduce(self) -> None:
        world = dist.get_world_size()
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / world)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n


def shard_range(total: int, rank: int, world: int):
    """Contiguous equal shard [lo, hi) of `total` samples (equal shards keep mean-of-shard-means == global mean)."""
    if total % world:
        raise ValueError(f"global batch {total} is not divisible by world size {world}")
    per = total // world
    return rank * per, (rank + 1) * per


def max_over_ranks(seconds: float, device) -> float:
    """The step time of the job is the slowest rank's (bench.py contract)."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
