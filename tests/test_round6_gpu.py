"""Round-6 parity tests on the GPU: teacher-forced bars on the AUGMENTED IMAGE at every full size (VERDICT r5 missing 2 / next 3) - the image bar of record.

The free-running image comparisons (tests/test_round3_gpu.py::test_full_size_*, tests/test_round4_gpu.py, tests/test_round5_gpu.py::test_shipped_workload_vs_reference_run)
measure one draw of a chaotic K-step loop and carry draw-calibrated bars (up to 1e-2 max norm at the shipped shapes).  This file binds the image itself, without any
calibration: north_star's 1e-4 of the image range in max norm, 1e-5 rms, at two points of the reference's own fp64 trajectory of every full-size call, both conv forms."""
import pytest
import torch

import r6_cases as R6
from parity_util import set_engine_default

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("winograd", [True, False], ids=["winograd", "direct"])
@pytest.mark.parametrize("point", ["initial", "final"])
@pytest.mark.parametrize("call", R6.CALLS)
def test_augmented_image_teacher_forced(dev, monkeypatch, call, point, winograd):
    """apply_max_style (encoder_decoder.py:598-631; MaxStyle.forward maxstyle.py:157-188) through the drop-in solver, ONE decode (n_iter = 0: nothing chaotic), at the
    reference's fp64 parameters - injected (`initial`: the first forward computes the batch std) and after its K-th step with the frozen batch std (`final`: the image
    generate_max_style_image returns, advanced_triplet...py:565-571) - against the reference's fp64 image at that point.
    Bar (north_star): max norm <= 1e-4 of the image range, rms <= 1e-5; per-plane mean / rms of the whole image <= 2e-5.  No calibration on draws."""
    set_engine_default(monkeypatch, "winograd", winograd)
    r = R6.image_teacher_forced(dev, call, point)
    print(f"image teacher-forced {call} {point} {'winograd' if winograd else 'direct'}: applied {r['applied']} max {r['image_max']:.2e} rms {r['image_rms']:.2e}"
          + (f" plane mean {r['plane_mean']:.2e} plane rms {r['plane_rms']:.2e}" if "plane_mean" in r else ""))
    assert r["winograd"] == winograd
    assert r["image_max"] <= 1e-4, r
    assert r["image_rms"] <= 1e-5, r
    if "plane_mean" in r:
        assert r["plane_mean"] <= 2e-5 and r["plane_rms"] <= 2e-5, r


@pytest.mark.parametrize("which", ["c2", "c4"])
def test_calls_are_bit_reproducible(dev, which):
    """One inner step of the benchmarked call, three times on one solver (eager first issue, captured replays): every encoder / segmentor buffer bit for bit - no
    kernel of the step depends on timing.  Round 6: the flat Winograd form (config 4's 20 x 20 levels) waited for its weights' LDS-DMA with a count that included
    halo loads the compiler had removed; one work item in ~10 calls multiplied with weights that had not landed (1e-5 of the range, the kink census and the
    teacher-forced bars failed intermittently).  Static twin: tests/test_isa_checks.py."""
    import r5_cases as R5
    runs = [R5.one_step_buffers(dev, which, True) for _ in range(3)]
    for r in runs[1:]:
        bad = [k for k in runs[0] if k in r and r[k].shape == runs[0][k].shape and not k.endswith(".stats") and not torch.equal(r[k], runs[0][k])]
        assert not bad, bad[:8]
