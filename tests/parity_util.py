"""Shared parity helpers (oracle tests on CPU, HIP-path tests on the GPU box).

Why "teacher forcing": the gradient of -CE w.r.t. the style parameters runs through ~60 conv/BN layers with
batch statistics and is badly conditioned - a 3e-8 relative change of the weights moves it by ~7e-4 *in fp64*
(measured while building the oracle), and Adam's first step is ~lr*sign(g), so a free-running K-step
trajectory amplifies fp32 rounding chaotically (the reference's own fp32 run differs from its fp64 run by
1.7e-4..2e-3 on the final image).  So every step is checked *from the reference's own
parameters at that step*: forward quantities tightly, gradients against a tolerance calibrated by the
reference's fp32-vs-fp64 gradient noise, and the parameter update (Adam) exactly given the reference's gradients.
"""
import numpy as np
import torch


def rel(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def style_names(layers):
    out = []
    for i in layers:
        out += [f"{i}.gamma_noise", f"{i}.beta_noise", f"{i}.lmda"]
    return out


def ref_params_at(g, step, layers, initial):
    """Reference style parameters after `step` Adam updates (step 0 = the injected initial state)."""
    if step == 0:
        return {n: initial[n] for n in style_names(layers)}
    return {n: g[f"step{step}.param.{n}"] for n in style_names(layers)}


def set_engine_default(monkeypatch, field, value):
    """Scoped change of an EngineOptions DEFAULT (maxstyle_amd/options.py) for engines built inside helpers that take no options argument: what
    `MS_OPTIONS=engine.<field>=<value>` does for a whole process, undone by monkeypatch at the end of the test."""
    from maxstyle_amd import options as O
    assert field in O._FIELDS, field
    monkeypatch.setitem(O._engine_defaults, field, value)
