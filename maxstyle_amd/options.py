"""Explicit configuration of the engines and of the library (round 5).

Until round 4 the product read 65 `MS_*` environment variables - 33 of them with getenv() inside the library's dispatch.  Now:
  * the LIBRARY reads nothing from the environment; where more than one kernel form is built for a shape it consults the option table of
    include/maxstyle_hip.h (`ms_set_option`; `set_library_option` / `library_options` here);
  * an ENGINE takes an `EngineOptions` object (or a dict of its fields); `AdvancedTripletReconSegmentationModel.loop_options / .train_options` hand one to the engines
    the solver builds.  Every field is an A/B switch between a fused launch and the launch sequence it replaces (bit-identical unless its comment says otherwise): the
    defaults are the product; the other settings keep the "same bits" tests and the A/B timing tools honest;
  * ONE environment variable is left for harnesses that cannot pass objects (tools/test_switches.sh runs the whole GPU suite under one changed option):
        MS_OPTIONS="engine.ride=0,conv.wino=0"
    parsed once at import by this module: `engine.<field>` entries change the DEFAULTS of EngineOptions, every other entry is a library option.
    (MS_LIB - an alternative build of the library, maxstyle_amd/_lib.py - and MS_SHARED_DEVICE - several ranks share this GPU, a deployment fact - are the other two
    variables the package reads.)"""
from __future__ import annotations

import dataclasses
import os
from dataclasses import dataclass
from typing import Optional


@dataclass
class EngineOptions:
    # Winograd F(2x2,3x3) form of the wide 3x3 stride-1 convolutions (MS_FETCH_WINOGRAD): on for the inner loop (its parity tests hold either way) and - since round 5 -
    # for the forward / data-gradient convs of the training passes (the weight-gradient kernels are their own).  Until round 4 the passes kept the direct form on a
    # one-batch measurement ("the form's rounding error on these activations is about twice the direct one's": more LeakyReLU kinks behind a backward pass).  The 20-seed
    # distribution of round 5 (tools/train_fidelity.py, profiles/r05_train_fidelity.json: weight gradients of the full-size pass against the fp64 oracle) shows no
    # difference: worst tensor per seed in L2, median / max over the seeds - direct 2.8e-3 / 8.3e-3, Winograd 2.5e-3 / 8.0e-3, the fp32 CPU oracle itself 2.4e-3 / 7.3e-3;
    # mean over tensors 7.1e-4 / 6.2e-4 / 5.2e-4.  -0.85 ms per trainer iteration (45.4 -> 47.3 iterations/s).
    winograd: Optional[bool] = None            # the inner loop's engines (and the module-forward engines); None = True
    train_winograd: Optional[bool] = None      # the training passes' engines (TrainEngine); None = True
    wino_appendix: bool = True        # the transformed weights are packed once per weight version behind the taps (MS_FETCH_WINO_U) instead of being recomputed per work item.
                                      # Acts where the weights are PACKED (engine.ConvW): the process default at that moment counts (MS_OPTIONS / engine_defaults), not an engine's own `options=`
    shared_device: Optional[bool] = None      # other kernels run beside the engine's launches: no co-residency kernels (single-read MaxStyle, `_xfin`).  None: MS_SHARED_DEVICE
    fuse_act_bwd: bool = True         # activation backward + BatchNorm-backward sums in the epilogue of the data-gradient conv (ms_conv2d_actbwd) vs a separate ms_act_bwd_reduce pass
    fuse_skip: bool = True            # residual-block tail as one launch (ms_conv1x1_bnres) vs ms_conv2d(ks=1) + ms_bn_act
    subpix: bool = True               # sub-pixel form of the two x2 resampling convolutions (results agree to fp32 rounding with the fused-fetch form)
    small_cout: bool = True           # vector-ALU kernel for the data-gradient that reaches the image
    small_cin: bool = True            # ms_conv3x3_small_cin for the encoder's first conv (taps-as-K matrix form: ms_conv2d's bits on rows of whole 16-pixel tiles)
    lazy_inc: bool = True             # the activation after the encoder's first double conv is never written (inner loop only)
    fuse_tail: bool = True            # ms_step_tail: the six launches that end a step as one
    fuse_fin_act: bool = True         # ms_bn_finalize_act for z_i / z_s (inner loop only)
    fuse_head_bwd: bool = True        # ms_style_bwd_head: layer 4's backward forms the head's input gradient itself
    ride: bool = True                 # ms_bn_bwd_coefs / ms_bn_finalize jobs ride on a residual block's 1x1 skip launch (ms_conv2d_ride)
    pool_fuse: bool = True            # producers of a masked gradient store its 2x2 sums themselves (agrees to rounding: grouping of the BatchNorm-backward sums)
    pool_epi: bool = True             # pooled Winograd epilogue (MS_EPI_POOL2) for the data-gradient of conv3x3(nearest-up-sampled x)
    lazy_style_head: bool = True      # the MaxStyle layer in front of the image head is never written (ms_head_fwd_styled)
    lazy_seg_tail: bool = True        # the segmentation decoder's last block output is never written (ms_head_ce_tail)
    xfin: bool = True                 # cross-workgroup BatchNorm finalize inside the consuming launch (`_xfin` entry points; needs an exclusive device)
    xfin_pro: bool = True             # ... also for consumers that need the coefficients in their prologue
    train_xfin: bool = False          # `_xfin` in the training engine's forward passes (-0.1 ms of 23: off)
    train_graph: bool = False         # training passes replayed from captured HIP graphs (same wall as eager: off)

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


_FIELDS = {f.name: f for f in dataclasses.fields(EngineOptions)}
_engine_defaults: dict = {}
_library_from_env: dict = {}


def _parse_value(v: str):
    v = v.strip()
    if v.lower() in ("none", ""):
        return None
    if v.lower() in ("true", "false"):
        return v.lower() == "true"
    return int(v)


def _parse_env(text: str):
    for item in text.split(","):
        item = item.strip()
        if not item:
            continue
        if "=" not in item:
            raise ValueError(f"MS_OPTIONS: '{item}' is not name=value")
        name, val = item.split("=", 1)
        name = name.strip()
        if name.startswith("engine."):
            f = name[len("engine."):]
            if f not in _FIELDS:
                raise ValueError(f"MS_OPTIONS: unknown engine option '{f}' (fields: {sorted(_FIELDS)})")
            pv = _parse_value(val)
            _engine_defaults[f] = None if pv is None else bool(pv)
        else:
            if name.startswith("diag."):
                raise ValueError(f"MS_OPTIONS: '{name}' is a timing-only ablation option (results are WRONG with it set): not accepted from the environment - a diagnostic "
                                 "tool sets it in its own process (set_library_option; tools/replay_conv.py <which> <reps> <dbg bits>)")
            _library_from_env[name] = int(val)


_parse_env(os.environ.get("MS_OPTIONS", ""))

# The 65 MS_* switches of rounds 1-4 are gone (round 5): a leftover in the environment would now be ignored without a word - MS_ACT_DTYPE=bf16 would run an fp32 loop,
# MS_TRAIN_WINOGRAD=0 a Winograd training pass.  Say so once at import.
_KNOWN_ENV = {"MS_OPTIONS", "MS_LIB", "MS_SHARED_DEVICE", "MS_SWITCH_MATRIX", "MS_MATRIX_EXTRA", "MS_TRACE_DUMP", "MS_R5_THREADS", "MS_GUARD_PAGES"}
_legacy = sorted(k for k in os.environ if k.startswith("MS_") and k not in _KNOWN_ENV)
if _legacy:
    import warnings
    warnings.warn("maxstyle_amd ignores the environment variable(s) " + ", ".join(_legacy) + ": the MS_* switches were replaced by MS_OPTIONS=\"engine.<field>=v,<library option>=v\", "
                  "EngineOptions objects and ms_set_option (maxstyle_amd/options.py); activation storage / bf16 arithmetic are solver attributes (loop_act_dtype, loop_mfma_dtype)",
                  RuntimeWarning, stacklevel=2)


def engine_options(given=None, **overrides) -> EngineOptions:
    """EngineOptions from: the class defaults, then the process defaults (MS_OPTIONS' `engine.*` entries, engine_defaults / set_engine_default), then `given`, then keyword
    overrides.  `given` as a DICT names the fields to change on top of the process defaults; `given` as an EngineOptions OBJECT is taken verbatim - every field of it, the
    process defaults do not reach such an engine (a caller who builds the object owns all of it; tools/test_switches.sh therefore reaches only engines built from dicts /
    defaults, which is every engine the solver builds unless loop_options / train_options hold an object)."""
    if isinstance(given, EngineOptions):
        opt = dataclasses.replace(given)
    else:
        opt = EngineOptions(**_engine_defaults)
        for k, v in (given or {}).items():
            if k not in _FIELDS:
                raise KeyError(f"unknown engine option '{k}' (fields: {sorted(_FIELDS)})")
            setattr(opt, k, v)
    for k, v in overrides.items():
        if k not in _FIELDS:
            raise KeyError(f"unknown engine option '{k}'")
        setattr(opt, k, v)
    if opt.shared_device is None:
        opt.shared_device = os.environ.get("MS_SHARED_DEVICE", "0") != "0"
    return opt


class engine_defaults:
    """with engine_defaults(winograd=False): ...   engines built inside take these DEFAULTS (what MS_OPTIONS' engine.* entries do for a process); for code that
    builds its engines inside helpers (bench.py's legs, the parity cases).  A caller that builds an engine itself passes `options=` instead."""

    def __init__(self, **kw):
        for k in kw:
            if k not in _FIELDS:
                raise KeyError(f"unknown engine option '{k}'")
        self.kw = kw

    def __enter__(self):
        self.saved = dict(_engine_defaults)
        _engine_defaults.update(self.kw)
        return self

    def __exit__(self, *exc):
        _engine_defaults.clear()
        _engine_defaults.update(self.saved)
        return False


def set_library_option(name: str, value: int) -> int:
    """ms_set_option: returns the previous value; raises on an unknown name or a value out of range."""
    from ._lib import lib, MaxStyleHipError
    prev = lib.ms_set_option(name.encode(), int(value))
    if prev < 0:
        raise MaxStyleHipError(lib.ms_last_error().decode(errors="replace"))
    return prev


def get_library_option(name: str) -> int:
    from ._lib import lib, MaxStyleHipError
    v = lib.ms_get_option(name.encode())
    if v < 0:
        raise MaxStyleHipError(lib.ms_last_error().decode(errors="replace"))
    return v


def library_options() -> dict:
    from ._lib import lib
    return {lib.ms_option_name(i).decode(): get_library_option(lib.ms_option_name(i).decode()) for i in range(lib.ms_option_count())}


class library_option:
    """with library_option("conv.k1s", 0): ...   (restores the previous value)"""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = set_library_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_library_option(self.name, self.prev)
        return False


def apply_env_library_options():
    """Called once by maxstyle_amd._lib after the library is loaded."""
    for name, val in _library_from_env.items():
        set_library_option(name, val)
