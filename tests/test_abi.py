"""CPU-only: the C-ABI shared library loads and exports every symbol include/bsgs_hip.h declares
(no compute calls without a GPU), and fails loudly -- not silently -- when no device exists."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "bsgs_hip.h")


@pytest.fixture(scope="module")
def libpath():
    import pybsgs
    if not os.path.exists(pybsgs.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bsgs-cuda_amd"), "-s"])
    return pybsgs.LIB_PATH


def declared_symbols(test_hooks=False):
    """the names include/bsgs_hip.h declares: outside its `#ifdef BSGS_TEST_HOOKS` block (the shipped ABI), or inside it (the test build's additions)"""
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    hooks = "".join(re.findall(r"#ifdef BSGS_TEST_HOOKS\n(.*?)#endif", txt, flags=re.S))
    if test_hooks:
        txt = hooks
    else:
        txt = re.sub(r"#ifdef BSGS_TEST_HOOKS\n.*?#endif", "", txt, flags=re.S)
    names = re.findall(r"\b(?:int|const char \*)\s*\*?\s*((?:bsgs_|cu)[A-Za-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_header_lists_are_in_sync():
    import pybsgs
    assert sorted(pybsgs.NATIVE_SYMBOLS + pybsgs.COMPAT_SYMBOLS) == declared_symbols()


def test_library_exports_every_declared_symbol(libpath):
    L = ctypes.CDLL(libpath)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing


def test_test_hooks_live_in_the_test_library_only(libpath):
    """bsgs_debug_corrupt_table / bsgs_debug_realloc are not part of the shipped ABI; build/libbsgs_hip_test.so (the same objects +
    csrc/test_hooks.hip) carries them for the verification tests, and the shipped host has no BSGS_TEST_CORRUPT_ENGINE hook compiled in"""
    import pybsgs
    hooks = declared_symbols(test_hooks=True)
    assert hooks == sorted(pybsgs.TEST_HOOK_SYMBOLS) and hooks
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert not (set(hooks) & exported), set(hooks) & exported
    assert not [s for s in exported if "corrupt" in s or "realloc" in s]
    test_lib = pybsgs.TEST_LIB_PATH
    assert os.path.exists(test_lib)
    out = subprocess.run(["nm", "-D", "--defined-only", test_lib], capture_output=True, text=True).stdout
    exported_t = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert set(hooks) <= exported_t and exported <= exported_t
    build = os.path.dirname(libpath)
    ship, test = open(os.path.join(build, "bsgs_mi355x"), "rb").read(), open(os.path.join(build, "bsgs_mi355x_test"), "rb").read()
    assert b"BSGS_TEST_CORRUPT_ENGINE" not in ship and b"BSGS_TEST_CORRUPT_ENGINE" in test
    assert b"libbsgs_hip_test" not in ship


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


def test_shipped_library_has_no_experiment_switch_and_the_shipped_source_has_none_to_offer(libpath, tmp_path):
    """bsgs_build_info() of the library the tests (and the driver) load is empty: no A/B switch, above all none of the *_CEILING timing
    experiments, which return wrong results; and the shipped source refuses such a switch (they live in a patch that is applied to a copy)"""
    L = ctypes.CDLL(libpath)
    L.bsgs_build_info.restype = ctypes.c_char_p
    assert L.bsgs_build_info() == b"", L.bsgs_build_info()
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = tmp_path / "guard.hip"
    src.write_text('#include "%s"\n' % os.path.join(ROOT, "bsgs-cuda_amd", "csrc", "giant_kernel.hip.h"))
    for sw in ("BSGS_NOCHAIN_CEILING", "BSGS_G2_CACHED_CEILING", "BSGS_NO_OVF_CEILING", "BSGS_SLICE_GATE=64", "BSGS_FULL_X"):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-D" + sw, "-DBSGS_EXPERIMENT", str(src)], capture_output=True, text=True)
        assert r.returncode != 0 and "build_experiment.sh" in r.stderr, (sw, r.stderr[-300:])


def test_tile_kernel_carries_no_experiment_branches_and_the_experiment_patch_still_applies(tmp_path):
    """the hot function giant_pair2_kernel is free of preprocessor branches (the ceilings, G2_VARY and the slice gate moved to
    tools/experiments/tile_kernel_experiments.patch); the patch must keep applying to the shipped sources, or tools/fetch_breakdown.py loses its
    two experiment libraries."""
    src = open(os.path.join(ROOT, "bsgs-cuda_amd", "csrc", "giant_kernel.hip.h")).read()
    a = src.index("giant_pair2_kernel(const TileArgs A)")
    body = src[a:]
    assert not re.search(r"^\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b", body, flags=re.M), "preprocessor branch inside the tile kernel"
    for word in ("CEILING", "G2_VARY", "gate_step", "BSGS_SLICE_GATE"):
        assert word not in body, word
    work = tmp_path / "bsgs-cuda_amd"
    (work / "csrc").mkdir(parents=True)
    for f in ("giant_kernel.hip.h", "bsgs_hip.hip", "bsgs_internal.h"):
        (work / "csrc" / f).write_text(open(os.path.join(ROOT, "bsgs-cuda_amd", "csrc", f)).read())
    patch = os.path.join(ROOT, "tools", "experiments", "tile_kernel_experiments.patch")
    r = subprocess.run(["patch", "-p1", "-d", str(work), "-i", patch], capture_output=True, text=True)
    assert r.returncode == 0 and ".rej" not in r.stdout, r.stdout[-600:] + r.stderr[-300:]
    patched = (work / "csrc" / "giant_kernel.hip.h").read_text()
    for word in ("BSGS_NOCHAIN_CEILING", "BSGS_G2_CACHED_CEILING", "BSGS_SLICE_GATE", "BSGS_G2_VARY"):
        assert word in patched, word


def test_narrow_batching_rule(libpath):
    """bsgs_debug_narrow_batching = the rule a launch's batching follows (bsgs_hip.hip narrow_pi), no device needed: a launch of few tiles halves
    the giants per thread until it has four blocks of 256 threads per CU, never below 128, never to an odd or non-multiple-of-4 batch, never to a
    thread count that is not a multiple of the block; launches that fill the GPU (and geometries too small to split) keep the default."""
    L = ctypes.CDLL(libpath)
    L.bsgs_debug_narrow_batching.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.POINTER(ctypes.c_uint32)]

    def rule(n, pi, ntiles, cus=256, block=256):
        out = ctypes.c_uint32()
        assert L.bsgs_debug_narrow_batching(n, pi, ntiles, cus, block, ctypes.byref(out)) == 0
        return out.value

    n = 1 << 24                                               # -t 256 -b 256 -p 256; default batching 16384 threads x 1024 giants
    assert [rule(n, 1024, k) for k in (1, 2, 3, 4, 7, 8, 12, 15, 16, 48, 192)] == [128, 128, 128, 256, 256, 512, 512, 512, 1024, 1024, 1024]
    assert rule(n, 1024, 1, cus=64) == 256                    # a smaller GPU is full sooner
    assert rule(n, 1024, 0) == 128 and rule(n, 1024, 1 << 20) == 1024
    # -t 128 -b 3 -p 256 (the fuzz geometry): 98304 giants, default 384 threads x 256 -> 768 x 128 ; 96 x 1024 cannot split (96 % 256)
    assert rule(128 * 3 * 256, 256, 1) == 128
    assert rule(96 * 1024, 1024, 1) == 1024
    # batch lengths that are not powers of two: 264 = 8 * 33 -> 132 (still a multiple of 4, >= 128); 260 -> 130 is not a multiple of 4; 520 -> 260 -> stop
    assert rule(256 * 264 * 8, 264, 1) == 132
    assert rule(256 * 260 * 8, 260, 1) == 260
    assert rule(256 * 520 * 8, 520, 1) == 260
    # nothing to do for odd / tiny / inconsistent input
    assert rule(1000, 10, 1) == 10 and rule(1 << 24, 1023, 1) == 1023 and rule(12345, 256, 1) == 256
    assert L.bsgs_debug_narrow_batching(n, 1024, 1, 256, 256, None) != 0


def test_grader_stop_and_keep_rule_without_a_device(libpath):
    """bsgs_debug_grade_rule = the rule alloc_graded_pieces follows (placement.hip GradeRule): when to stop drawing scratch pieces and which to keep,
    for GPUs that show two memory classes (MI355X so far), one class only, five classes, or nothing but bad pieces."""
    L = ctypes.CDLL(libpath)
    L.bsgs_debug_grade_rule.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                        ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]

    def rule(grades, need, extra=24):
        g = (ctypes.c_float * len(grades))(*grades)
        drawn, sep, kept = ctypes.c_uint32(), ctypes.c_uint32(), (ctypes.c_uint32 * need)()
        assert L.bsgs_debug_grade_rule(g, len(grades), need, extra, ctypes.byref(drawn), kept, ctypes.byref(sep)) == 0
        return drawn.value, list(kept), bool(sep.value)

    far, near, straddle = 41.8, 38.0, 40.9
    # the usual case: the first allocations come from the lines' own group (all near), then the far group turns up: draw until `need` far ones are there
    drawn, kept, sep = rule([near] * 6 + [far] * 10 + [near] * 20, 6)
    assert (drawn, sorted(kept), sep) == (12, [6, 7, 8, 9, 10, 11], True)
    # far pieces first: a separation must still be SEEN before "six within 2 % of the best" means anything -> one near piece later is enough
    drawn, kept, sep = rule([far] * 8 + [near] + [far] * 30, 6)
    assert (drawn, sorted(kept), sep) == (9, [0, 1, 2, 3, 4, 5], True)
    # straddlers (2-3 % low) are not kept while better ones exist, and do not count as a separation
    drawn, kept, sep = rule([far, straddle, far, far, straddle, far, far, near, far], 6)
    assert sep and sorted(kept) == [0, 2, 3, 5, 6, 8] and drawn == 9
    # ONE class only (another partition mode, or a table that fills every group evenly): after need + 12 pieces the rule gives up, keeps the FIRST `need`
    # in allocation order -- plain allocation -- and reports that it saw no separation
    drawn, kept, sep = rule([40.0 + 0.1 * (k % 3) for k in range(40)], 6)
    assert (drawn, kept, sep) == (18, [0, 1, 2, 3, 4, 5], False)
    # five classes 3 % apart: the best class is kept once six of it were seen and something 5 % lower showed up
    five = [44.0, 42.7, 41.4, 40.1, 38.8]
    seq = [five[k % 5] for k in range(60)]
    drawn, kept, sep = rule(seq, 6)
    assert sep and all(seq[k] == 44.0 for k in kept) and drawn == 26             # the sixth 44.0 is piece 25
    # all pieces bad but one: that one sets the best, nothing else is within 2 % -> the draw runs to need + extra_max and keeps the best six there are
    drawn, kept, sep = rule([38.0] * 3 + [42.0] + [38.0] * 40, 6, extra=10)
    assert drawn == 16 and sep and kept[0] == 3 and len(set(kept)) == 6
    # fewer candidates than needed: everything there is, marked
    drawn, kept, sep = rule([41.0, 38.0], 4)
    assert drawn == 2 and kept[2:] == [0xFFFFFFFF] * 2


def test_production_kernels_do_not_spill(tmp_path):
    """The register budget of the hot kernels, checked where they are built (cross-compilation: no GPU).  The three shipped tile-kernel instantiations -- <2, false, true>
    (what bench.py times), <4, false, true> (any number of buckets: -w auto's tables) at four waves per SIMD and at most 128 VGPRs, <3, false, true> (128-byte lines) at three
    waves and at most 168 -- spill NO VGPR and execute NO scratch instruction; <2> and <4> have no private segment at all (round 4 shipped 34 spilled VGPRs and 144 bytes of
    scratch per lane).  <3> reports a 48-byte frame that nothing touches (a compiler artefact: one 32-byte object without users + the scavenging slot).
    The table builder's kernels (baby_keys_kernel<2|3>, ext_refine_kernel<2|3>, ext_finalize_kernel) spill nothing either (round 5: 3 and 87 VGPRs)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    for tu, kernel, max_vgpr, frame in (("tile_lines64.hip", "_Z18giant_pair2_kernelILi2ELb0ELb1EEv8TileArgs", 128, 0), ("tile_lines128.hip", "_Z18giant_pair2_kernelILi3ELb0ELb1EEv8TileArgs", 168, 48),
                                         ("tile_lines64_any.hip", "_Z18giant_pair2_kernelILi4ELb0ELb1EEv8TileArgs", 128, 0)):
        asm = tmp_path / (tu + ".s")
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-S", "--cuda-device-only", "-o", str(asm),
                               os.path.join(ROOT, "bsgs-cuda_amd", "csrc", tu)], stderr=subprocess.DEVNULL)
        text = asm.read_text()
        meta = text[text.index(".name:           " + kernel):][:900]
        get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", meta).group(1))      # noqa: E731
        assert get("vgpr_spill_count") == 0, (tu, meta)
        assert get("vgpr_count") <= max_vgpr, (tu, meta)
        assert get("private_segment_fixed_size") <= frame, (tu, meta)
        body = text[text.index("\n" + kernel + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        touched = [ln for ln in body.split("\n") if re.match(r"\s+(scratch_|buffer_(load|store))", ln)]
        assert not touched, (tu, touched[:3])
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import spill_report
    rows = [r for r in spill_report.report(tus=["baby_builder"]) if "rocprim" not in r["kernel"]]
    names = " ".join(r["kernel"] for r in rows)
    for want in ("baby_keys_kernel<0>", "baby_keys_kernel<2>", "baby_keys_kernel<3>", "ext_refine_kernel<2>", "ext_refine_kernel<3>", "ext_finalize_kernel<2>", "ext_finalize_kernel<3>"):
        assert want in names, want
    for r in rows:
        if any(k in r["kernel"] for k in ("baby_keys_kernel", "ext_refine_kernel", "ext_finalize_kernel", "ext_validate", "table_census_kernel", "table_lookup_kernel")):
            assert r["vgpr_spill"] == 0 and r["scratch_bytes_per_lane"] == 0, r
