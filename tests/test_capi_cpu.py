"""CPU: the C-ABI library loads, exports every symbol of include/gto_solver.h, agrees with the
oracle on the default options and fails loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from grasptrajopt_amd import _capi
    return _capi


def test_library_exports_all_declared_symbols(capi):
    hdr = open(os.path.join(ROOT, "include", "gto_solver.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gto_[a-z_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTED_SYMBOLS)
    lib = capi.load_library()
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_default_opts_match_reference_constants(capi, oracle_mod):
    o = capi.default_opts()
    r = oracle_mod.reference_opts()
    for name, _ in capi.CSolverOpts._fields_:
        assert getattr(o, name) == getattr(r, name), name
    assert (o.T, o.Tmax, o.standoff_offset, o.w_obstacle, o.w_vel, o.max_iter) == (50, 10.0, -10, 10.0, 0.01, 100)
    hdr = open(os.path.join(ROOT, "include", "gto_solver.h")).read()
    abi = int(re.search(r"#define GTO_ABI_VERSION (\d+)", hdr).group(1))
    assert capi.load_library().gto_version() == abi == capi.ABI_VERSION


def test_library_of_another_abi_is_refused(capi, tmp_path):
    """A library named by GTO_HIP_LIB for an A/B run has to speak this wrapper's ABI: one whose gto_version() says otherwise is
    refused when it is loaded, not called with the signatures of another header."""
    import subprocess
    src = tmp_path / "old.c"
    src.write_text("int gto_version(void) { return 1000; }\n")
    so = tmp_path / "libgto_old.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)])
    with pytest.raises(RuntimeError, match="ABI version 1000"):
        capi.load_library(str(so))


def test_create_validates_and_has_no_cpu_fallback(capi):
    import torch
    from grasptrajopt_amd.robot_desc import load_builtin
    d = load_builtin("panda")
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-device path cannot be exercised")
    with pytest.raises(capi.GTOError, match="no HIP device|fallback"):
        capi.SolverHandle(d, "panda_hand", "panda_hand")
    # argument validation happens before any device work
    bad = capi.default_opts()
    bad.T = 2
    with pytest.raises(capi.GTOError, match="T must be"):
        capi.SolverHandle(d, "panda_hand", "panda_hand", bad)
    bad = capi.default_opts()
    bad.standoff_offset = -60
    with pytest.raises(capi.GTOError, match="standoff"):
        capi.SolverHandle(d, "panda_hand", "panda_hand", bad)


def test_missing_library_fails_loudly(capi, tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load_library(str(tmp_path / "libgto_hip.so"))
