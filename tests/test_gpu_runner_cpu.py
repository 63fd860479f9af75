"""The per-file child-process runner of tests/conftest.py, exercised on the CPU with dummy "gpu" tests: results
are replayed one to one, a child that aborts costs exactly the test it was running (reported with the child's
output), and the tests after it still run -- in a fresh child."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

DUMMY = '''
import os, sys
import pytest
pytestmark = pytest.mark.gpu

def test_a_passes():
    assert os.environ.get("MSR3D_GPU_CHILD")          # runs in a child

def test_b_fails():
    assert 1 == 2, "expected failure text"

def test_c_skips():
    pytest.skip("no such shape")

def test_d_aborts():
    os.write(2, b"Memory access fault by GPU node-1 (dummy)\\n")       # as the HSA runtime does: straight to fd 2
    os.abort()

@pytest.mark.parametrize("k", [0, 1])
def test_e_after_the_abort(k):
    assert k in (0, 1)
'''

OTHER = '''
import pytest
pytestmark = pytest.mark.gpu

def test_other_file():
    pass
'''


def _run(tmp_path, *args):
    tests = tmp_path / "tests"
    tests.mkdir(exist_ok=True)
    shutil.copy(os.path.join(HERE, "conftest.py"), tests / "conftest.py")
    (tests / "test_dummy_gpu.py").write_text(DUMMY)
    (tests / "test_other_gpu.py").write_text(OTHER)
    env = dict(os.environ, MSR3D_GPU_ASSUME="1")
    env.pop("MSR3D_GPU_CHILD", None)
    env.pop("MSR3D_GPU_INPROC", None)
    return subprocess.run([sys.executable, "-m", "pytest", "tests/", "-q", "-m", "gpu", "-p", "no:cacheprovider", *args],
                          cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)


def test_child_per_file_replays_results_and_survives_an_abort(tmp_path):
    r = _run(tmp_path)
    out = r.stdout + r.stderr
    assert r.returncode == 1, out
    assert "4 passed" in out and "2 failed" in out and "1 skipped" in out, out
    assert "expected failure text" in out
    assert "Memory access fault by GPU node-1 (dummy)" in out          # the dying child's own words are kept
    assert "SIGABRT" in out
    assert "test_d_aborts" in out


def test_maxfail_stops_at_the_first_failure(tmp_path):
    r = _run(tmp_path, "-x")
    out = r.stdout + r.stderr
    assert r.returncode == 1, out
    assert "1 passed" in out and "1 failed" in out and "skipped" not in out, out
