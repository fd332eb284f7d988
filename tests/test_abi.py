"""CPU-only: the C-ABI shared library loads and exports every symbol include/bsgs_hip.h declares
(no compute calls without a GPU), and fails loudly -- not silently -- when no device exists."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "bsgs_hip.h")


@pytest.fixture(scope="module")
def libpath():
    import pybsgs
    if not os.path.exists(pybsgs.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bsgs-cuda_amd"), "-s"])
    return pybsgs.LIB_PATH


def declared_symbols():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(?:int|const char \*)\s*\*?\s*((?:bsgs_|cu)[A-Za-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_header_lists_are_in_sync():
    import pybsgs
    assert sorted(pybsgs.NATIVE_SYMBOLS + pybsgs.COMPAT_SYMBOLS) == declared_symbols()


def test_library_exports_every_declared_symbol(libpath):
    L = ctypes.CDLL(libpath)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing


def test_no_silent_cpu_fallback(libpath):
    """Without a GPU the product path must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pybsgs
    with pytest.raises(pybsgs.BsgsError):
        pybsgs.Device(0)


def test_product_does_not_link_or_import_the_oracle(libpath):
    out = subprocess.run(["ldd", libpath], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bsgs-cuda_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".inc")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in src and "oracle_lib" not in src and "curve64_ref" not in src, f
