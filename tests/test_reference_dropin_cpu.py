"""Drop-in at the registry boundary, shown with the REFERENCE's own code:
the reference's `OSE3DSituation` is built through the reference's `modules.build` registry with
OUR `PcdObjEncoder` registered under the same name, loads the same state dict, and reproduces
the golden `obj_tokens` the all-reference model produced; and the reference's
`model.build.build_model` instantiates OUR `OSE3DSituation` from the same config.  Runs only where /root/reference is
mounted (the build container); in a subprocess, because importing the reference installs stub
modules (fvcore, easydict, ...) process-wide."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from tests.golden.make_golden import import_reference, ref_cfg
from tests.helpers import fill_state_dict, load_golden, golden_inputs, rel_l2
pu, ose = import_reference()
import importlib
ref_build = importlib.import_module("modules.build")            # the REFERENCE's registries
ref_enc = importlib.import_module("modules.vision.pcd_pointnet_encoder")
assert ref_build.VISION_REGISTRY.get("PcdObjEncoder") is ref_enc.PcdObjEncoder

# our encoder, with the oracle standing in for the HIP ops on this GPU-less host
from oracle import pn2
from msr3d_amd.pointnet2 import pointnet2_utils as our_pu
our_pu._ext = pn2.ext_module()
from msr3d_amd.modules.vision.pcd_pointnet_encoder import PcdObjEncoder as OurEncoder
ref_build.VISION_REGISTRY.t["PcdObjEncoder"] = OurEncoder        # INTEGRATION.md section 2

for variant, st in (("transform", "as_transform_for_objects"), ("anchor", "as_object")):
    g = load_golden(variant, 0)
    model = ose.OSE3DSituation(ref_cfg(st)).eval()               # reference model, our encoder inside
    assert type(model.obj_encoder) is OurEncoder
    assert sorted(model.state_dict().keys()) == list(g["state_keys"])
    model.load_state_dict(fill_state_dict(model.state_dict(), 0), strict=True)
    with torch.no_grad():
        out = model(golden_inputs(g))
    err = rel_l2(out["obj_tokens"].numpy(), g["obj_tokens"])
    assert err < 2e-5, (variant, err)
    assert np.array_equal(out["obj_masks"].numpy(), g["obj_masks_out"])
    print(variant, "reference OSE3DSituation + our PcdObjEncoder: rel-L2", err)

# the other direction: the reference's `model.build.build_model` instantiating OUR OSE3DSituation
ref_model_build = importlib.import_module("model.build")
from msr3d_amd.model.ose3d_situation import OSE3DSituation as OurModel
ref_model_build.MODEL_REGISTRY.t["OSE3DSituation"] = OurModel
g = load_golden("transform", 1)
ours = ref_model_build.build_model(ref_cfg("as_transform_for_objects")).eval()
assert type(ours) is OurModel and hasattr(ours, "obj_encoder") and hasattr(ours, "device")
ours.load_state_dict(fill_state_dict(ours.state_dict(), 1), strict=True)
with torch.no_grad():
    out = ours(golden_inputs(g))
err = rel_l2(out["obj_tokens"].numpy(), g["obj_tokens"])
assert err < 2e-5, err
print("reference build_model -> our OSE3DSituation: rel-L2", err)
print("DROPIN_OK")
'''


@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="reference checkout not mounted")
def test_reference_model_builds_and_runs_with_our_encoder_registered():
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "DROPIN_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
