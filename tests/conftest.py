"""pytest plumbing.

* `gpu` marker; GPU tests are skipped (not failed) when no device is visible.
* **One child process per GPU test FILE.**  A GPU fault (`Memory access fault by GPU ...`), an abort inside a
  vendor library or a hung kernel takes the whole interpreter with it; run in-process under `-x` it also takes
  every file collected after it (round 4's driver run: one SIGABRT in `test_pn2_modules_gpu.py`, 286 tests in 15
  files never reached).  So when a device is visible the parent pytest process runs no GPU test itself: the
  first GPU item of a file starts `python -m pytest <that file's selected node ids>` as a child whose conftest
  (this file, `MSR3D_GPU_CHILD` set) appends one JSON line per finished test phase to a results file, and the
  parent replays those lines as its own test reports -- so `-q`, `-x`, the pass count and failure text read as
  usual.  If the child dies, the test it was running is reported FAILED with the child's exit status and the
  tail of its output (the message a fault prints before `abort()` is kept, not cut), and the rest of the file
  runs in a fresh child.  Nothing is retried.  `MSR3D_GPU_INPROC=1` restores the single-process run (profilers
  that must see the kernels in the process they launched).
"""
import json
import os
import signal
import subprocess
import sys
import tempfile
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_CHILD_ENV = "MSR3D_GPU_CHILD"           # path of the results file: set in a child
_FILE_TIMEOUT = float(os.environ.get("MSR3D_GPU_FILE_TIMEOUT", "1500"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    if os.environ.get("MSR3D_GPU_ASSUME") == "1":       # tests/test_gpu_runner_cpu.py: the runner itself, on dummies
        return True
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ------------------------------------------------------------------------------------------ child side
def pytest_runtest_logreport(report):
    path = os.environ.get(_CHILD_ENV)
    if not path:
        return
    text = ""
    if report.outcome != "passed":
        lr = report.longrepr
        text = lr[2] if (report.outcome == "skipped" and isinstance(lr, tuple) and len(lr) == 3) else str(lr)
    rec = {"nodeid": report.nodeid, "when": report.when, "outcome": report.outcome, "text": text,
           "duration": getattr(report, "duration", 0.0), "xfail": getattr(report, "wasxfail", None)}
    with open(path, "a") as f:
        f.write(json.dumps(rec) + "\n")
        f.flush()
        os.fsync(f.fileno())


# ----------------------------------------------------------------------------------------- parent side
class _FileRun:
    """The selected GPU items of one test file, run in child processes; results by node id."""

    def __init__(self, config, nodeids):
        self.config = config
        self.pending = list(nodeids)     # not yet attempted by any child
        self.results = {}                # nodeid -> {"setup": rec, "call": rec, "teardown": rec}
        self.died = {}                   # nodeid -> text (the child died while running it)

    def _spawn(self, nodeids):
        fd, res_path = tempfile.mkstemp(prefix="msr3d_gpu_", suffix=".jsonl")
        os.close(fd)
        log_fd, log_path = tempfile.mkstemp(prefix="msr3d_gpu_", suffix=".log")
        env = dict(os.environ)
        env[_CHILD_ENV] = res_path
        env.setdefault("PYTHONFAULTHANDLER", "1")
        # --capture=sys: what C code (the HSA runtime's fault message, a library's abort text) writes to fd 2
        # reaches the log instead of dying with the child's fd-level capture file
        cmd = [sys.executable, "-m", "pytest", *nodeids, "-q", "-m", "gpu", "-p", "no:cacheprovider",
               "--rootdir", ROOT, "--capture=sys"]
        if self.config.getoption("maxfail", 0) == 1:
            cmd.append("-x")
        t0 = time.time()
        proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=log_fd, stderr=subprocess.STDOUT,
                                start_new_session=True)
        os.close(log_fd)
        note = ""
        try:
            rc = proc.wait(timeout=_FILE_TIMEOUT)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            rc = proc.wait()
            note = f"killed after {_FILE_TIMEOUT:.0f} s (MSR3D_GPU_FILE_TIMEOUT)\n"
        recs = []
        with open(res_path) as f:
            for line in f:
                try:
                    recs.append(json.loads(line))
                except ValueError:
                    pass                 # a line cut by the death of the child
        with open(log_path, errors="replace") as f:
            tail = f.read()[-6000:]
        os.unlink(res_path)
        os.unlink(log_path)
        return rc, recs, note + tail, time.time() - t0

    def _run_some(self):
        nodeids, self.pending = self.pending, []
        rc, recs, tail, secs = self._spawn(nodeids)
        for r in recs:
            self.results.setdefault(r["nodeid"], {})[r["when"]] = r
        complete = lambda n: "teardown" in self.results.get(n, {})
        rest = [n for n in nodeids if not complete(n)]
        if not rest:
            return
        stopped_on_failure = rc == 1 and any(
            ph["outcome"] == "failed" for n in nodeids for ph in self.results.get(n, {}).values())
        if stopped_on_failure:           # the child's own -x: the parent stops at the same report
            return
        sig = f" (signal {-rc}: {signal.Signals(-rc).name})" if rc < 0 else ""
        self.died[rest[0]] = (f"the child pytest process running this test ended with status {rc}{sig} after "
                              f"{secs:.0f} s before the test finished; tail of its output:\n{tail}")
        self.pending = rest[1:]          # the rest of the file: a fresh child, when asked for

    def result(self, nodeid):
        while nodeid not in self.results and nodeid not in self.died and self.pending:
            self._run_some()
        if nodeid in self.died:
            return None, self.died[nodeid]
        phases = self.results.get(nodeid)
        if phases is None:
            return None, "the child pytest process did not report this test (deselected or not collected there?)"
        return phases, None


_runs = {}


def _delegating(config):
    return (not os.environ.get(_CHILD_ENV) and os.environ.get("MSR3D_GPU_INPROC", "0") != "1" and _has_gpu())


def pytest_collection_finish(session):
    if not _delegating(session.config):
        return
    by_file = {}
    for item in session.items:
        if "gpu" in item.keywords:
            by_file.setdefault(str(item.fspath), []).append(item.nodeid)
    for path, nodeids in by_file.items():
        _runs[path] = _FileRun(session.config, nodeids)


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_protocol(item, nextitem):
    run = _runs.get(str(item.fspath))
    if run is None or "gpu" not in item.keywords:
        return None
    from _pytest.reports import TestReport
    hook = item.ihook
    hook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    phases, died = run.result(item.nodeid)

    def emit(when, outcome, longrepr, duration=0.0, xfail=None):
        rep = TestReport(nodeid=item.nodeid, location=item.location, keywords=dict(item.keywords), outcome=outcome,
                         longrepr=longrepr, when=when, duration=duration)
        if xfail is not None:
            rep.wasxfail = xfail
        hook.pytest_runtest_logreport(report=rep)

    if phases is None:
        emit("setup", "passed", None)
        emit("call", "failed", died)
        emit("teardown", "passed", None)
    else:
        for when in ("setup", "call", "teardown"):
            r = phases.get(when)
            if r is None:
                continue
            longrepr = None
            if r["outcome"] == "skipped":
                longrepr = (str(item.fspath), item.location[1], r["text"])
            elif r["outcome"] == "failed":
                longrepr = r["text"]
            emit(when, r["outcome"], longrepr, r.get("duration", 0.0), r.get("xfail"))
    hook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)
    return True
