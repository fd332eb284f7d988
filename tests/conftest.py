import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_fx():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "small_w1024_ht8_t2_b2_p4.json")) as f:
        return json.load(f)


def free_hbm(want_bytes, wait_s=20.0):
    """free memory of GPU 0, giving a previous test's process up to `wait_s` to hand its memory back (a 190 GiB table is released -- and wiped by the driver --
    asynchronously: a test that follows a big one at once would see a GPU that is still 'full' and skip for no reason)"""
    import time
    import torch
    t0 = time.time()
    while True:
        free = torch.cuda.mem_get_info(0)[0]
        if free >= want_bytes or time.time() - t0 > wait_s:
            return free
        time.sleep(0.5)


def rerun_in_test_library(request, timeout=1800):
    """Tests of the VERIFICATION need a corrupted table, and the hook that corrupts one (bsgs_debug_corrupt_table) lives in build/libbsgs_hip_test.so only
    -- the shipped objects plus csrc/test_hooks.hip -- never in the library the other tests (and the driver) load.  Such a test starts with
        if rerun_in_test_library(request): return
    which runs this very test in a child pytest whose pybsgs loads the test library (BSGS_LIB_PATH) and asserts that it passed there; inside the child it returns
    False and the body runs."""
    import subprocess
    import pybsgs
    if os.environ.get("BSGS_LIB_PATH") == pybsgs.TEST_LIB_PATH:
        return False
    assert os.path.exists(pybsgs.TEST_LIB_PATH), "make -C bsgs-cuda_amd builds build/libbsgs_hip_test.so"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", request.node.nodeid], cwd=ROOT, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, BSGS_LIB_PATH=pybsgs.TEST_LIB_PATH))
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    return True
