"""ctypes binding of libmaxstyle_hip.so (the C ABI declared in include/maxstyle_hip.h).

There is deliberately NO fallback: if the shared library is missing the import fails loudly, and every op
refuses non-CUDA tensors.  Build with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C maxstyle_amd/csrc`."""
import ctypes
import os

import torch  # noqa: F401  MUST precede the CDLL below: PyTorch-ROCm ships its own libamdhip64; loading ours after it makes the
               # dynamic loader reuse that one runtime (same SONAME). Loaded the other way round the process ends up with two HIP
               # runtimes and ours reports 'no ROCm-capable device' on the first launch.

_HERE = os.path.dirname(os.path.abspath(__file__))
# MS_LIB: an alternative build of the same library (A/B timing of compile-time choices: `make BUILD=.. OUT=.. EXTRA=-D..` in csrc/); default = the in-tree build
LIB_PATH = os.environ.get("MS_LIB") or os.path.join(_HERE, "lib", "libmaxstyle_hip.so")

c_f32p = ctypes.c_void_p   # raw device pointers travel as integers
c_i64p = ctypes.c_void_p
c_void = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size = ctypes.c_size_t

class TailLayer(ctypes.Structure):
    """`ms_tail_layer` of include/maxstyle_hip.h (ms_step_tail)."""
    _fields_ = [("part", ctypes.c_void_p), ("mu", ctypes.c_void_p), ("sig", ctypes.c_void_p), ("gamma_std", ctypes.c_void_p), ("beta_std", ctypes.c_void_p),
                ("perm", ctypes.c_void_p), ("off_gamma", c_int), ("off_beta", c_int), ("off_lmda", c_int), ("learn_noise", c_int), ("learn_mix", c_int),
                ("B", c_int), ("C", c_int), ("S", c_int)]


# name -> (restype, argtypes): mirrors include/maxstyle_hip.h one to one (tests/test_abi.py checks both ways)
SIGNATURES = {
    "ms_version": (c_int, []),
    "ms_last_error": (ctypes.c_char_p, []),
    "ms_set_option": (c_int, [ctypes.c_char_p, c_int]),
    "ms_get_option": (c_int, [ctypes.c_char_p]),
    "ms_option_default": (c_int, [ctypes.c_char_p]),
    "ms_option_count": (c_int, []),
    "ms_option_name": (ctypes.c_char_p, [c_int]),
    "ms_diag_set_trace": (c_int, [c_void, c_void]),
    "ms_diag_k3n_chain_bytes": (c_size, [c_int]),
    "ms_diag_k3n_chain": (c_int, [c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_void, c_void, c_void, c_void]),
    "ms_style_ws_bytes": (c_size, [c_int, c_int, c_int]),
    "ms_style_moments": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_float, c_void, c_size, c_void]),
    "ms_style_coeffs": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_int, c_int, c_void]),
    "ms_style_apply": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_void]),
    "ms_style_fwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p,
                             c_int, c_int, c_int, c_float, c_void, c_size, c_void]),
    "ms_style_fused_ws_bytes": (c_size, [c_int, c_int, c_int]),
    "ms_num_cus": (c_int, []),
    "ms_style_fused_plan": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "ms_style_ws_state_offset": (c_size, [c_int, c_int, c_int]),
    "ms_style_fused_status": (c_int, [c_void, ctypes.POINTER(c_int), c_void]),
    "ms_style_fwd_fused": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p,
                                   c_int, c_int, c_int, c_float, c_void, c_size, c_void]),
    "ms_style_fwd_3k": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p,
                                c_int, c_int, c_int, c_float, c_void, c_size, c_void]),
    "ms_pool2_actbwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_float, c_void]),
    "ms_add_actbwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_float, c_f32p, c_void]),
    "ms_conv2d_pool2_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "ms_conv2d_form": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "ms_wino_pack_floats": (c_size, [c_int, c_int]),
    "ms_wino_pack": (c_int, [c_f32p, c_int, c_int, c_void]),
    "ms_pool2_actbwd_pool": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_float, c_f32p, c_void]),
    "ms_head_ce_actbwd_parts": (c_int, [c_int, c_int, c_int]),
    "ms_head_ce_actbwd": (c_int, [c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_void, c_int, c_int, c_int, c_int, c_float, c_void, c_size,
                                  c_f32p, c_f32p, c_f32p, c_float, c_void]),
    "ms_head_ce_tail": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_void, c_int, c_int, c_int, c_int, c_int, c_float, c_void, c_size,
                                c_f32p, c_float, c_f32p, c_void]),
    "ms_style_bwd_actbwd_parts": (c_int, [c_int, c_int, c_int]),
    "ms_style_bwd_actbwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p,
                                    c_int, c_int, c_int, c_void, c_size, c_f32p, c_f32p, c_f32p, c_float, c_void]),
    "ms_style_bwd_head": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p,
                                  c_int, c_int, c_int, c_void, c_size, c_f32p, c_f32p, c_f32p, c_float, c_void]),
    "ms_style_ws_bytes_bf16": (c_size, [c_int, c_int, c_int]),
    "ms_style_fused_ws_bytes_bf16": (c_size, [c_int, c_int, c_int]),
    "ms_style_fwd_bf16": (c_int, [c_void, c_void, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p,
                                  c_int, c_int, c_int, c_float, c_void, c_size, c_void]),
    "ms_style_fwd_fused_bf16": (c_int, [c_void, c_void, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p,
                                        c_int, c_int, c_int, c_float, c_void, c_size, c_void]),
    "ms_style_bwd_bf16": (c_int, [c_void, c_void, c_void, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p,
                                  c_int, c_int, c_int, c_void, c_size, c_void]),
    "ms_style_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p,
                             c_int, c_int, c_int, c_void, c_size, c_void]),
    "ms_adam_step": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_float, c_float, c_float, c_float, c_int, c_void, c_void]),
    "ms_counter_incr": (c_int, [c_void, c_void]),
    "ms_style_bwd_slots": (c_int, [c_int, c_int, c_int, c_int]),
    "ms_step_tail": (c_int, [ctypes.POINTER(TailLayer), c_int, c_void, c_int, ctypes.c_double, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                             c_float, c_float, c_float, c_float, c_void, c_void, c_void]),
    "ms_conv_stats_bytes": (c_size, [c_int, c_int, c_int, c_int]),
    "ms_conv_stats_parts": (c_int, [c_int, c_int, c_int]),
    "ms_conv2d": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                          c_int, c_f32p, c_f32p, c_f32p, c_int, c_int, c_float, c_int, c_f32p, c_void]),
    "ms_conv_k1s_would_run": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "ms_conv_ride_capacity": (c_int, [c_int, c_int, c_int]),
    "ms_conv2d_ride": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_f32p, c_f32p, c_f32p, c_int, c_int, c_float, c_int, c_f32p, c_int, c_f32p, c_int, c_f32p, c_f32p, c_float, ctypes.c_double, c_f32p, c_int, c_void]),
    "ms_bn_finalize": (c_int, [c_f32p, c_int, c_f32p, c_f32p, c_float, c_f32p, c_int, c_void]),
    "ms_bn_act": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_int, c_int, c_int, c_int, c_float, c_void]),
    "ms_bn_finalize_act": (c_int, [c_f32p, c_int, c_f32p, c_f32p, c_float, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_float, c_void]),
    "ms_act_bwd_parts": (c_int, [c_int, c_int, c_int]),
    "ms_act_bwd_reduce": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_float, c_void]),
    "ms_conv1x1_bnres": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_float, c_int, c_void]),
    "ms_xfin_gran_bytes": (c_size, [c_int]),
    "ms_conv1x1_bnres_xfin": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_void, c_void,
                                      c_float, c_int, c_void]),
    "ms_conv2d_xfin": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_f32p,
                               c_int, c_f32p, c_f32p, c_f32p, c_float, ctypes.c_double, c_f32p, c_void, c_void, c_void]),
    "ms_conv2d_actbwd_xfin": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_float, c_f32p,
                                      c_f32p, c_f32p, ctypes.c_double, c_f32p, c_void, c_void, c_void]),
    "ms_conv3x3_small_cout_ok": (c_int, [c_int, c_int]),
    "ms_conv3x3_small_cin_ok": (c_int, [c_int, c_int, c_int]),
    "ms_conv3x3_small_cin": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_f32p, c_void]),
    "ms_conv3x3_small_cout": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_int, c_void]),
    "ms_conv_subpix_eligible": (c_int, [c_int, c_int]),
    "ms_conv_subpix": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_void]),
    "ms_conv_subpix2": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_int, c_void]),
    "ms_subpix_pack_floats": (c_size, [c_int, c_int]),
    "ms_subpix_pack": (c_int, [c_f32p, c_f32p, c_int, c_int, c_void]),
    "ms_appendix_desc_bytes": (c_size, []),
    "ms_appendix_job_threads": (ctypes.c_longlong, [c_int, c_int, c_int]),
    "ms_appendix_batch": (c_int, [c_void, c_int, ctypes.c_longlong, c_void]),
    "ms_conv_actbwd_tab_bytes": (ctypes.c_size_t, [c_int]),
    "ms_conv2d_actbwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_f32p, c_f32p, c_f32p, c_int, c_int, ctypes.c_float, c_f32p, c_f32p, ctypes.c_float, c_f32p, c_void]),
    "ms_clock_probe": (c_int, [c_int, c_int, c_int, c_void, c_void, c_void]),
    "ms_bn_bwd_coefs": (c_int, [c_f32p, c_int, c_f32p, ctypes.c_double, c_f32p, c_int, c_void]),
    "ms_pool2_sum": (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_void]),
    "ms_conv_wgrad_ws_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "ms_conv_wgrad": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_int, c_float, c_int, c_void, c_size, c_void]),
    "ms_bn_bwd_full": (c_int, [c_f32p, c_int, c_f32p, ctypes.c_double, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_void]),
    "ms_channel_sum_ws_bytes": (c_size, [c_int, c_int]),
    "ms_channel_sum": (c_int, [c_f32p, c_int, c_int, c_int, c_f32p, c_int, c_void, c_size, c_void]),
    "ms_head_wgrad_ws_bytes": (c_size, [c_int, c_int, c_int, c_int]),
    "ms_head_wgrad": (c_int, [c_f32p, c_f32p, c_void, c_int, c_float, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_void, c_size, c_void]),
    "ms_head_wgrad_ds": (c_int, [c_f32p, c_f32p, c_void, c_int, c_float, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_void, c_size, c_void]),
    "ms_mse_loss_ds": (c_int, [c_f32p, c_f32p, c_size, c_float, c_float, c_f32p, c_f32p, c_f32p, c_void, c_size, c_void]),
    "ms_head_ce_ds": (c_int, [c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p, c_void, c_int, c_int, c_int, c_int, c_float, c_f32p,
                              c_void, c_size, c_void]),
    "ms_mse_ws_bytes": (c_size, []),
    "ms_mse_loss": (c_int, [c_f32p, c_f32p, c_size, c_float, c_float, c_f32p, c_f32p, c_void, c_size, c_void]),
    "ms_adamw_step": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_size, c_float, c_float, c_float, c_float, c_float, c_int, c_void, c_void]),
    "ms_repack_desc_bytes": (c_size, []),
    "ms_repack_weights": (c_int, [c_f32p, c_void, c_int, ctypes.c_longlong, c_void]),
    "ms_bn_running_update": (c_int, [c_f32p, c_f32p, c_f32p, c_int, ctypes.c_double, c_float, c_float, c_void]),
    "ms_conv_wgrad_partials": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_int, c_float, c_void, c_size, ctypes.POINTER(c_int), c_void]),
    "ms_wgrad_batch_desc_bytes": (c_size, []),
    "ms_wgrad_reduce_batch": (c_int, [c_void, c_int, ctypes.c_longlong, c_void]),
    "ms_bn_running_desc_bytes": (c_size, []),
    "ms_bn_running_update_batch": (c_int, [c_void, c_int, c_float, c_float, c_void]),
    "ms_rescale_intensity": (c_int, [c_f32p, c_f32p, c_int, c_int, c_float, c_float, c_float, c_void]),
    "ms_confusion": (c_int, [c_f32p, c_i64p, c_void, c_int, c_int, c_int, c_void]),
    "ms_head_fwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_void]),
    "ms_head_fwd_styled": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_void]),
    "ms_head_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_void]),
    "ms_head_ce_ws_bytes": (c_size, [c_int, c_int]),
    "ms_head_ce": (c_int, [c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p, c_void, c_int, c_int, c_int, c_int, c_float,
                           c_void, c_size, c_void]),
}
# bf16 activation storage for the conv stack: same argument lists as the fp32 entry points (device pointers travel as integers either way)
for _n in ("ms_conv2d", "ms_conv2d_ride", "ms_conv2d_xfin", "ms_conv2d_actbwd_xfin", "ms_conv1x1_bnres", "ms_conv1x1_bnres_xfin", "ms_conv2d_actbwd", "ms_bn_act", "ms_bn_finalize_act", "ms_act_bwd_reduce", "ms_pool2_sum", "ms_pool2_actbwd", "ms_pool2_actbwd_pool", "ms_add_actbwd",
           "ms_head_fwd", "ms_head_fwd_styled", "ms_head_bwd", "ms_head_ce", "ms_head_ce_actbwd", "ms_head_ce_tail", "ms_conv_subpix", "ms_conv3x3_small_cout", "ms_conv3x3_small_cin", "ms_style_bwd_actbwd", "ms_style_bwd_actbwd_parts"):
    SIGNATURES[_n + "_bf16"] = SIGNATURES[_n]
for _n in ("ms_conv2d", "ms_conv2d_actbwd"):           # bf16 storage + bf16 matrix arithmetic (v_mfma_f32_16x16x16_bf16) where built
    SIGNATURES[_n + "_bf16m"] = SIGNATURES[_n]


class MaxStyleHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MaxStyle HIP extension is not built. "
            "Run `make -C maxstyle_amd/csrc` (or __graft_entry__.build()). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
from . import options as _options  # noqa: E402
_options.apply_env_library_options()      # MS_OPTIONS' library entries (harness hook; the library itself reads no environment)


def check(status, what):
    if status != 0:
        msg = lib.ms_last_error().decode(errors="replace")
        raise MaxStyleHipError(f"{what} failed with status {status}: {msg}")
