"""The floating-point contract of the index ops' squared distance is a switch, in the oracle (run time) and in
the kernels (build time; msr3d_sqdist_contract() reports it), and tools/fma_contract_risk.py measures what the
choice changes (/root/reference/modules/third_party/pointnet2/_ext_src/src/
sampling_gpu.cu:99-104, ball_query_gpu.cu:32-35)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_the_contracts_are_different_roundings_of_the_same_sum():
    from oracle import pn2
    L = pn2.lib()
    L.pn2o_sq3.restype = ctypes.c_float
    L.pn2o_sq3.argtypes = [ctypes.c_float] * 3
    rng = np.random.default_rng(0)
    differ = 0
    try:
        for _ in range(2000):
            a, b, c = rng.standard_normal(3).astype(np.float32)
            vals = []
            for k in range(4):
                pn2.set_contract(k)
                vals.append(L.pn2o_sq3(a, b, c))
            exact = float(a) ** 2 + float(b) ** 2 + float(c) ** 2
            assert all(abs(v - exact) <= 2.0 ** -22 * exact for v in vals)        # each within ~2 ulp of the sum
            differ += len(set(vals)) > 1
        # contract 0 is the chain the kernels spell out: fma(c, c, fma(a, a, b * b))
        pn2.set_contract(0)
        a, b, c = np.float32(0.1), np.float32(0.7), np.float32(-0.3)
        want = np.float32(np.float64(c) * np.float64(c) + np.float64(np.float32(np.float64(a) * np.float64(a) + np.float64(b * b))))
        assert L.pn2o_sq3(a, b, c) == want
    finally:
        pn2.set_contract(0)
    assert differ > 400          # the choice is visible in the distances themselves (~40 % of triples) ...


def test_the_library_is_built_with_the_default_contract():
    from msr3d_amd import _lib
    from oracle import pn2
    assert _lib.load().msr3d_sqdist_contract() == 0 == pn2.get_contract()


def test_risk_report_on_a_small_sample():
    """... but not in the decisions: the report's schema, and that on 2 bench scenes + the tie-heavy clouds no
    index vector changes under any contraction (the full-size numbers are in profiles/r03_fma_contract_risk.json)."""
    import fma_contract_risk as fr
    res = fr.run(scenes=2, test_b=8, encoder=True, random_clouds=64)
    assert res["clouds"] == 120 and set(res["contracts"]) == {"1", "2", "3", "4", "5"}
    for c, e in res["contracts"].items():
        for k, v in e.items():
            if isinstance(v, dict):
                assert 0.0 <= v["clouds_differing"] <= 1.0 and 0.0 <= v["entries_differing"] <= 1.0
        if c in ("1", "2", "3"):
            assert e["bench/fps1"]["clouds_differing"] <= 0.01 and e["bench/ball1"]["entries_differing"] <= 1e-5
            assert e["enc_out_rel_l2"] <= 1e-3
