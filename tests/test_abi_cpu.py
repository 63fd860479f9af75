"""CPU: the C-ABI library loads and exports every symbol include/msr3d_hip.h declares;
argument validation works without a device (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    from msr3d_amd import _lib
    h = _lib.load()
    header = open(os.path.join(ROOT, "include", "msr3d_hip.h")).read()
    declared = set(re.findall(r"\b(msr3d_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(h, sym), f"{sym} declared in the header but not exported"
    assert declared == set(_lib.exported_symbols())
    assert h.msr3d_abi_version() == int(re.search(r"#define MSR3D_ABI_VERSION (\d+)", header).group(1)) >= 2


def test_invalid_arguments_are_rejected_without_a_device():
    from msr3d_amd import _lib
    h = _lib.load()
    null = ctypes.c_void_p(0)
    # n <= 0 / null pointers -> MSR3D_EINVAL, never a crash or an exit()
    assert h.msr3d_furthest_point_sampling(1, 0, 4, null, null, null, null) == -22
    assert h.msr3d_furthest_point_sampling(1, 8, 4, null, null, null, null) == -22
    assert h.msr3d_furthest_point_sampling(0, 8, 4, null, null, null, null) == 0      # empty batch
    assert h.msr3d_ball_query(-1, 8, 4, ctypes.c_float(0.2), 4, null, null, null, null) == -22
    assert h.msr3d_group_points(1, 1, 8, 2, 2, null, null, null, null) == -22
    assert h.msr3d_status_string(-22) == b"invalid argument"
    assert h.msr3d_status_string(0) == b"ok"
    # empty problems are a successful no-op
    assert h.msr3d_gather_points(0, 3, 8, 4, null, null, null, null) == 0


def test_no_oracle_import_in_product():
    """The product package must never import, link or call the oracle."""
    pkg = os.path.join(ROOT, "msr3d_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libpn2_oracle" not in src and "pn2o_" not in src, f
