"""Guard pages for the GPU tests: every tensor the package allocates ends (or starts) at the edge of its own hipMalloc'd block.

On this pool a hipMalloc of a multiple of 2 MiB is followed by an unmapped page (tools/probes/oob_probe.hip: one byte past the end is a "Memory access fault by GPU",
the process aborts); smaller blocks are carved from mapped 2 MiB fragments and a read past their end returns silently.  With PYTORCH_NO_CUDA_MEMORY_CACHING=1 every
torch allocation is its own hipMalloc, so a tensor placed at the END of a block rounded up to 2 MiB turns any access past its last element into a crash - a past-the-end
sanitizer for the kernels' operand addressing (GPU AddressSanitizer is not available here).  Round 6 found such a read by accident (the Winograd appendix, DESIGN.md
section 3); this makes the search deliberate.

    MS_GUARD_PAGES=end|start PYTORCH_NO_CUDA_MEMORY_CACHING=1 python -m pytest tests/... -m gpu        (tests/conftest.py installs it; tests/test_guard_pages_gpu.py runs it)

What is guarded: every torch.empty / zeros / ones / full / *_like call made from inside maxstyle_amd (engine buffers, op outputs, packed weights with their appendix,
tables, workspaces) and every host-to-device `Tensor.to` copy (the tests' inputs).  `end`: the tensor's last byte is within 15 bytes of the block's end (16-byte
alignment of the start is kept - the kernels' vector paths check it); `start`: the tensor starts the block (a 2 MiB-aligned address whose preceding page belongs to
no allocation of this process)."""
import importlib
import os
import pkgutil

import torch

TWO_MB = 2 << 20
MODE = os.environ.get("MS_GUARD_PAGES", "")


def _numel(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


def _shape_of(args):
    if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
        return tuple(int(s) for s in args[0])
    return tuple(int(s) for s in args)


def guarded_empty(shape, dtype=None, device=None, mode=None):
    mode = mode or MODE or "end"
    dtype = dtype or torch.get_default_dtype()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    n = _numel(shape)
    if dev.type != "cuda" or n == 0:
        return torch.empty(shape, dtype=dtype, device=dev)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    total = (nbytes + 15 + TWO_MB - 1) // TWO_MB * TWO_MB
    big = torch.empty(total, dtype=torch.uint8, device=dev)
    off = ((total - nbytes) & ~15) if mode == "end" else 0
    return big[off:off + nbytes].view(dtype).view(shape)


def guarded(t, mode=None):
    """A copy of the device tensor `t` in a guarded block."""
    g = guarded_empty(tuple(t.shape), t.dtype, t.device, mode)
    g.copy_(t)
    return g


class _TorchProxy:
    """Stands in for the `torch` module inside maxstyle_amd's modules: allocation calls are guarded, everything else is torch's."""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def _split(args, kw):
        kw = dict(kw)
        dtype, device = kw.pop("dtype", None), kw.pop("device", None)
        if kw:                                      # (requires_grad, pin_memory, out=...: torch's own call)
            return None
        return _shape_of(args), dtype, device

    def empty(self, *args, **kw):
        s = self._split(args, kw)
        return torch.empty(*args, **kw) if s is None else guarded_empty(*s)

    def zeros(self, *args, **kw):
        s = self._split(args, kw)
        return torch.zeros(*args, **kw) if s is None else guarded_empty(*s).zero_()

    def ones(self, *args, **kw):
        s = self._split(args, kw)
        return torch.ones(*args, **kw) if s is None else guarded_empty(*s).fill_(1)

    def full(self, size, fill_value, **kw):
        s = self._split((size,), kw)
        return torch.full(size, fill_value, **kw) if s is None else guarded_empty(*s).fill_(fill_value)

    def empty_like(self, t, **kw):
        return guarded_empty(tuple(t.shape), kw.get("dtype", t.dtype), kw.get("device", t.device))

    def zeros_like(self, t, **kw):
        return self.empty_like(t, **kw).zero_()


_orig_to = torch.Tensor.to


def _guarded_to(self, *args, **kw):
    """Tensor.to under the guard: a fresh host-to-device copy (the tests' `_rand(...).to(dev)` inputs) lands in a guarded block too."""
    r = _orig_to(self, *args, **kw)
    if r.is_cuda and not self.is_cuda and r.numel() > 0 and r.is_contiguous() and not r.requires_grad and type(r) is torch.Tensor:
        return guarded(r)
    return r


def install():
    """Swap the `torch` name inside every maxstyle_amd module for the proxy, and Tensor.to for the guarded copy (idempotent)."""
    torch.Tensor.to = _guarded_to
    if os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") != "1":
        raise RuntimeError("MS_GUARD_PAGES needs PYTORCH_NO_CUDA_MEMORY_CACHING=1 (every allocation its own hipMalloc)")
    import maxstyle_amd
    proxy = _TorchProxy()
    n = 0
    for m in pkgutil.iter_modules(maxstyle_amd.__path__):
        mod = importlib.import_module("maxstyle_amd." + m.name)
        if getattr(mod, "torch", None) is torch:
            mod.torch = proxy
            n += 1
    return n
