"""GPU: the earlier forms of the scene blocks that the environment still selects (MSR3D_ATTN_FWD_SPLIT=0: one workgroup per
(scene, head); MSR3D_ATTN_FWD_WAVES=4; MSR3D_ATTN_BWD=2 / 0: the four-wave backward kernels; MSR3D_FFN_WAVES=4;
MSR3D_FC_SPLIT=0) stay under the same checker as the defaults: the blocks-vs-strips comparison of every intermediate and
gradient at 2 scenes x 60 objects with dropout, and the encoder against the reference-generated fixture.  The switches are read
once per process, so each form runs in a process of its own."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FORMS = [{"MSR3D_ATTN_FWD_SPLIT": "0"}, {"MSR3D_ATTN_FWD_SPLIT": "0", "MSR3D_ATTN_FWD_WAVES": "4"}, {"MSR3D_ATTN_BWD": "2"},
         {"MSR3D_ATTN_BWD": "0"}, {"MSR3D_FFN_WAVES": "4"}]


@pytest.mark.parametrize("form", FORMS, ids=lambda f: ",".join(f"{k[6:]}={v}" for k, v in f.items()))
def test_earlier_block_forms_match_the_strips_schedule(form):
    env = dict(os.environ, MSR3D_GPU_INPROC="1", **form)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_scene_blocks_gpu.py"), "-q", "-x",
                          "-m", "gpu", "-k", "test_blocks_schedule_matches_strips and (2-60-512 or 3-37-256)"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "2 passed" in out.stdout, out.stdout[-3000:] + out.stderr[-1000:]


def test_fc_on_the_panel_kernel_matches_the_fixture():
    env = dict(os.environ, MSR3D_GPU_INPROC="1", MSR3D_FC_SPLIT="0")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_golden_fullsize_gpu.py"), "-q", "-x",
                          "-m", "gpu"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-1000:]
