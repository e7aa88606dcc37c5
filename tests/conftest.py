import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The two-rank job of tests/test_gpu_two_ranks.py: its ranks must be children of a process that has NOT initialised the GPU
# (a process that has may not start other programs on the GPU pool), and every GPU test of this session initialises it - so the
# job is started HERE, before collection (torch.cuda.device_count() does not initialise the device; torch.cuda.is_available()
# below does), runs beside the first tests, and the test only collects its record.
TWO_RANK_JOB = {"proc": None, "out": None}


def _wants_two_rank_job(config) -> bool:
    expr = config.getoption("-m", default="") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return False
    if config.getoption("-k", default=""):
        return "two_rank" in config.getoption("-k")
    files = [a for a in config.args if a.endswith(".py") or "::" in a]
    return not files or any("test_gpu_two_ranks" in a for a in files)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("PYTEST_XDIST_WORKER") or not _wants_two_rank_job(config):
        return
    import torch
    if torch.cuda.device_count() < 1:
        return
    out = os.path.join(tempfile.mkdtemp(prefix="two_rank_"), "result.json")
    TWO_RANK_JOB["out"] = out
    TWO_RANK_JOB["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_two_rank_step.py"), "launch", out],
                                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def pytest_unconfigure(config):
    proc = TWO_RANK_JOB["proc"]
    if proc is not None and proc.poll() is None:
        proc.terminate()                    # the exact PID started above (its ranks end with their rendezvous)
        try:
            proc.wait(timeout=20)
        except subprocess.TimeoutExpired:
            proc.kill()


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
