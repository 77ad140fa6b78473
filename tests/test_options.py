"""The configuration surface (round 5): the library reads nothing from the environment, its kernel-form choices are an explicit option table
(ms_set_option), the engines take an EngineOptions object, and ONE variable (MS_OPTIONS) is left as the harness hook.  CPU tests: no launch happens here."""
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_sources_never_read_the_environment():
    srcs = glob.glob(os.path.join(ROOT, "maxstyle_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "maxstyle_amd", "csrc", "*.hip")) + \
        glob.glob(os.path.join(ROOT, "maxstyle_amd", "csrc", "*.cpp"))
    assert len(srcs) > 20
    for f in srcs:
        text = re.sub(r"//[^\n]*", "", open(f).read())
        assert "getenv" not in text, f


def test_package_reads_three_environment_variables():
    """MS_LIB (an alternative build), MS_SHARED_DEVICE (several ranks share the GPU: a deployment fact), MS_OPTIONS (harness hook) - nothing else."""
    names = set()
    for f in glob.glob(os.path.join(ROOT, "maxstyle_amd", "*.py")):
        names |= set(re.findall(r"environ(?:\.get)?\(\s*[\"'](MS_[A-Z0-9_]+)[\"']", open(f).read()))
        names |= set(re.findall(r"environ\[\s*[\"'](MS_[A-Z0-9_]+)[\"']", open(f).read()))
    assert names == {"MS_LIB", "MS_SHARED_DEVICE", "MS_OPTIONS"}, names


def test_option_table_round_trip():
    from maxstyle_amd._lib import lib
    from maxstyle_amd import options as O
    n = lib.ms_option_count()
    names = [lib.ms_option_name(i).decode() for i in range(n)]
    assert 8 <= n <= 16 and len(set(names)) == n and lib.ms_option_name(n) is None
    for nm in names:
        assert O.get_library_option(nm) == lib.ms_option_default(nm.encode())
    with O.library_option("conv.k1s", 0):
        assert O.get_library_option("conv.k1s") == 0
    assert O.get_library_option("conv.k1s") == 1
    with pytest.raises(Exception):
        O.set_library_option("no.such.option", 1)
    with pytest.raises(Exception):
        O.set_library_option("conv.wino", 7)          # out of range
    assert lib.ms_set_option(None, 0) < 0


def test_engine_options_object():
    from maxstyle_amd import options as O
    d = O.engine_options()
    assert d.winograd is None and d.train_winograd is None and d.ride and d.xfin and d.small_cin and d.shared_device is False
    assert len(O._FIELDS) <= 24
    assert O.engine_options({"ride": False}).ride is False
    assert O.engine_options(O.EngineOptions(xfin=False)).xfin is False
    with pytest.raises(KeyError):
        O.engine_options({"overlap": True})           # a switch that was removed in round 5
    with O.engine_defaults(winograd=False):
        assert O.engine_options().winograd is False
    assert O.engine_options().winograd is None


def test_ms_options_harness_hook():
    code = ("from maxstyle_amd import options as O; import maxstyle_amd._lib; "
            "print(O.engine_options().ride, O.engine_options().train_winograd, O.get_library_option('conv.wino'), O.get_library_option('conv.wide_rows'))")
    env = dict(os.environ, MS_OPTIONS="engine.ride=0, engine.train_winograd=1,conv.wino=2,conv.wide_rows=8", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split()[-4:] == ["False", "True", "2", "8"], r.stdout
    bad = subprocess.run([sys.executable, "-c", code], env=dict(env, MS_OPTIONS="engine.nonsense=1"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "unknown engine option" in bad.stderr
