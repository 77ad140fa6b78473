"""Import harness for the *reference* implementation (only usable where /root/reference exists).

Used exclusively by tests/golden/make_golden.py (fixture generation in the build container) and by the
optional `tests/test_oracle_vs_reference.py` (skipped when /root/reference is absent, e.g. on the GPU box).
Nothing from the reference is copied: it is imported in place, with stub packages standing in for its
un-installed, path-irrelevant dependencies (plotting, medical IO, ...).
"""
import collections
import collections.abc
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MAXSTYLE_REFERENCE_ROOT", "/root/reference")

_STUB_ROOTS = (
    "tkinter", "torchvision", "monai", "torchio", "SimpleITK", "medpy", "skimage", "cv2", "seaborn",
    "umap", "IPython", "torchsample", "tensorboard", "tensorboardX", "advchain", "nibabel", "matplotlib",
)
_STUB_EXACT = ("numpy.lib.function_base", "scipy.misc", "numpy.core.fromnumeric")


class _Anything(types.ModuleType):
    """A module whose every attribute is another permissive stub (callable, subclassable)."""

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        obj = type(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})
        setattr(self, name, obj)
        return obj


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in _STUB_ROOTS or fullname in _STUB_EXACT:
            if fullname in sys.modules:
                return None
            if root not in _STUB_ROOTS:
                # only stub an exact dotted name if the real one is missing
                try:
                    for f in sys.meta_path:
                        if f is self:
                            continue
                        spec = f.find_spec(fullname, path, target)
                        if spec is not None:
                            return None
                except Exception:
                    pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Anything(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise FileNotFoundError(REFERENCE_ROOT)
    sys.meta_path.insert(0, _StubFinder())
    if not hasattr(collections, "MutableMapping"):
        collections.MutableMapping = collections.abc.MutableMapping
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src"))


def load_maxstyle_cls():
    install()
    from src.advanced.maxstyle import MaxStyle
    return MaxStyle


def load_solver_module():
    """Returns the reference solver module with `MaxStyle` rebound to force use_gpu=False
    (the reference hard-codes the CUDA default inside generate_max_style_image)."""
    install()
    import functools
    import src.models.advanced_triplet_recon_segmentation_model as solver_mod
    from src.advanced.maxstyle import MaxStyle

    class CpuMaxStyle(MaxStyle):
        created = []  # every instance built by the solver, in construction order

        def __init__(self, *a, **k):
            k["use_gpu"] = False
            super().__init__(*a, **k)
            hook = getattr(CpuMaxStyle, "post_init_hook", None)
            if hook is not None:
                hook(self, len(CpuMaxStyle.created))
            CpuMaxStyle.created.append(self)

    solver_mod.MaxStyle = CpuMaxStyle
    solver_mod.CpuMaxStyle = CpuMaxStyle
    return solver_mod
