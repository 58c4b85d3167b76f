"""CPU-side checks of the drop-in boundary: every function declared in include/gnark_amd.h is exported by the
hipcc-built shared library and bound by the Python mirror; no compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gnark_amd.h")
SO = os.path.join(ROOT, "gnark_amd", "libgnark_amd.so")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ga_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_so():
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "gnark_amd", "csrc"), "-j", str(os.cpu_count() or 4)],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return SO


def test_header_functions_all_bound_in_python():
    from gnark_amd import _lib
    assert set(declared_functions()) == set(_lib.EXPORTED_SYMBOLS)


def test_shared_library_exports_every_declared_symbol(built_so):
    dll = ctypes.CDLL(built_so)
    missing = [f for f in declared_functions() if not hasattr(dll, f)]
    assert not missing, missing
    from gnark_amd import _lib
    lib = _lib.Library(built_so)            # binds every prototype
    assert b"gfx950" in lib.ga_version()
    c, nw = ctypes.c_int(), ctypes.c_int()
    assert lib.ga_msm_plan(0, 0, 1 << 20, ctypes.byref(c), ctypes.byref(nw)) == 0   # host-only planner
    assert c.value >= 10 and nw.value == 254 // c.value + 1
    assert lib.ga_msm_plan(7, 0, 16, ctypes.byref(c), ctypes.byref(nw)) != 0        # unknown curve id -> error, message set
    assert b"curve" in lib.ga_last_error()


@pytest.mark.parametrize("curve", [0, 1], ids=["bn254", "bls12-381"])
@pytest.mark.parametrize("group", [0, 1], ids=["G1", "G2"])
def test_host_side_group_arithmetic_of_the_built_library(built_so, curve, group):
    """the HOST arithmetic of the hipcc-built library (64-bit-limb Montgomery products, field.hip.h `mul_host64` -- the emulation
    build keeps the device form, so this is the only CPU-side check of it): ga_msm_combine_windows, the Horner step of a raw MSM,
    on window sums [k_j]G against [sum_j 2^(c j) k_j]G from the C oracle.  No device is touched."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    import pyref
    from gnark_amd import _lib
    lib = _lib.Library(built_so)
    r = (pyref.BN254, pyref.BLS12_381)[curve].r
    rng = np.random.default_rng(77 + 2 * curve + group)
    for nwin, c in ((1, 22), (4, 13), (13, 20), (16, 16)):
        ks = [int(x) for x in rng.integers(1, 1 << 40, size=nwin)]
        W = np.ascontiguousarray(np.stack([oracle.generator_mul(curve, group, k) for k in ks]))
        out = np.zeros_like(W[0])
        assert lib.ga_msm_combine_windows(curve, group, W.ctypes.data_as(ctypes.c_void_p), nwin, c, out.ctypes.data_as(ctypes.c_void_p)) == 0
        want = oracle.generator_mul(curve, group, sum(k << (c * j) for j, k in enumerate(ks)) % r)
        assert np.array_equal(oracle.jac_to_affine(curve, group, out), oracle.jac_to_affine(curve, group, want)), (nwin, c)


def test_every_entry_point_is_an_exception_barrier():
    """every extern "C" definition of the library is a function-try-block ending in GA_ABI_CATCH (common.hip.h): the count of guarded
    bodies equals the count of definitions, file by file, and together they are every function the header declares except the two
    that return a constant string"""
    import re
    total = 0
    for f in ("abi.hip", "groth16.hip", "hash_to_field.hip"):
        src = open(os.path.join(ROOT, "gnark_amd", "csrc", f)).read()
        defs = re.findall(r'^(?:extern "C" )?(?:int|void) (ga_\w+)\([^;{]*\) (try )?\{', src, flags=re.M)
        assert defs and all(t for _, t in defs), [n for n, t in defs if not t]
        assert len(defs) == len(re.findall(r"^\} GA_ABI_CATCH(?:_VOID)?$", src, flags=re.M)), f
        assert src.count("GA_ABI_ENTRY();") == len(defs), f
        total += len(defs)
    assert total == len(declared_functions()) - 2   # ga_last_error, ga_version


def test_no_cpu_fallback_in_package():
    """The product package must not import the oracle or the emulation build."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gnark_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hip.h", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "import oracle" not in txt and "pyref" not in txt, f
                assert "libgnark_amd_emu" not in txt, f


def test_missing_library_fails_loudly(tmp_path):
    from gnark_amd import _lib
    with pytest.raises(_lib.GnarkAmdError, match="no CPU fallback"):
        _lib.Library(str(tmp_path / "nope.so"))


def test_plain_c_client_runs_against_the_emulation_build(tmp_path, emu_lib):
    """the C99 client of include/gnark_amd.h (tests/c_abi/abi_client.c: MSM, table MSM, NTT round trip, PLONK grand product,
    batch inversion, hash-to-field) compiled with gcc and linked against the emulation build of the same sources -- the header
    is plain C and the flows work without Python in between; the GPU suite runs the same program against the HIP build"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_client_emu")
    emu_dir = os.path.join(root, "tests", "emu")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-DABI_CLIENT_N=256", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", "abi_client.c"), os.path.join(emu_dir, "libgnark_amd_emu.so"),
                           "-Wl,-rpath," + emu_dir, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ABI_CLIENT_OK" in r.stdout, r.stdout + r.stderr


def test_cgo_call_pattern_and_concurrency_against_the_emulation_build(tmp_path, emu_lib):
    """tests/c_abi/cgo_pattern.c: the Go shim's call sequence replayed from C (key staged through ga_g16_builder_* with every
    source buffer poisoned after its call, struct-of-pointers variant with the struct in C heap, solutions poisoned after
    ga_g16_prove) and the concurrency guarantees of icicle.go:77-86,821-823 (two threads on one context interleaved with ga_fft,
    a second context on the same device) -- all proofs identical"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cgo_pattern_emu")
    emu_dir = os.path.join(root, "tests", "emu")
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-pthread", "-DCGO_LOGN=6", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", "cgo_pattern.c"), os.path.join(emu_dir, "libgnark_amd_emu.so"),
                           "-Wl,-rpath," + emu_dir, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "CGO_PATTERN_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("curve", ["GA_BN254", "GA_BLS12_381"])
def test_plonk_call_pattern_against_the_emulation_build(tmp_path, emu_lib, curve):
    """tests/c_abi/plonk_pattern.c: the device call sequence of the PLONK prover with prove.patch applied (pin both SRS, batched
    commitment of L, R, O, grand product, pinned + un-pinned quotient, batched commitment of h1..h3, the two openings incl. the fold),
    every buffer transient and poisoned after its call, replayed twice to identical outputs -- here against the emulation build"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "plonk_pattern_emu")
    emu_dir = os.path.join(root, "tests", "emu")
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-DPLONK_LOGN=6", "-DPLONK_CURVE=" + curve, "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", "plonk_pattern.c"), os.path.join(emu_dir, "libgnark_amd_emu.so"),
                           "-Wl,-rpath," + emu_dir, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PLONK_PATTERN_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("curve", ["bn254", "bls12-381"])
def test_plonk_prover_patch_applies_to_the_reference(tmp_path, curve):
    """go/backend/accelerated/mi355x/plonk/<curve>/prove.patch (the Accelerator hooks wired into backend/plonk/<curve>/prove.go)
    applies cleanly to the reference's file -- the call sites it names exist as quoted"""
    import shutil
    import subprocess
    ref = "/root/reference/backend/plonk/%s/prove.go" % curve
    if not os.path.exists(ref):
        pytest.skip("reference tree not present on this box")
    shutil.copy(ref, tmp_path / "prove.go")
    patch = os.path.join(ROOT, "go", "backend", "accelerated", "mi355x", "plonk", curve, "prove.patch")
    r = subprocess.run(["patch", "-p4", "--dry-run", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "FAILED" not in r.stdout, r.stdout + r.stderr
    src = open(patch).read()
    for hook in ("CommitLagrangeBatch", "CommitBatch", "BuildRatioCopyConstraint", "ComputeQuotientRaw", "s.acc.Open(", "LinearCombination"):
        assert hook in src, hook


def test_go_shim_identifiers_resolve():
    """tools/check_go_idents.py: every `pkg.Ident` of go/** is declared in the reference package it is imported from (or, for the
    gnark-crypto dependency that is not in the tree, used by the reference under the same import path), and every `C.ga_*` call
    matches a prototype of include/gnark_amd.h argument count included.  The Go side has never met a compiler (no toolchain in the
    image); this is the part of a build that can be done without one.  go/IDENTS.json is the committed result."""
    import json
    import subprocess
    import sys
    table = json.load(open(os.path.join(ROOT, "go", "IDENTS.json")))
    assert table["unresolved"] == [] and len(table["resolved"]) > 200
    if not os.path.isdir("/root/reference"):
        pytest.skip("reference tree not present on this box: the committed go/IDENTS.json stands")
    out = os.path.join(ROOT, "tests", "emu", "build", "idents_check.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_go_idents.py"), "--json", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    fresh = json.load(open(out))
    assert fresh["resolved"] == table["resolved"], "go/IDENTS.json is stale: run python tools/check_go_idents.py --json go/IDENTS.json"


@pytest.mark.parametrize("curve", ["bn254", "bls12-381"])
def test_groth16_randomness_hook_patch_applies_to_the_reference(tmp_path, curve):
    """go/backend/accelerated/mi355x/internal/fixtures/groth16_rs_<curve>.patch -- the test-only hook that lets gnark's own CPU prover
    take (r, s) and show its solution, so that gen_fixtures_test.go can write golden directories (tests/gnark_fixture.py) -- applies
    cleanly to backend/groth16/<curve>/prove.go: the lines it names (prove.go:105,171-177) exist as quoted"""
    import shutil
    import subprocess
    ref = "/root/reference/backend/groth16/%s/prove.go" % curve
    if not os.path.exists(ref):
        pytest.skip("reference tree not present on this box")
    dst = tmp_path / "backend" / "groth16" / curve
    dst.mkdir(parents=True)
    shutil.copy(ref, dst / "prove.go")
    fx = os.path.join(ROOT, "go", "backend", "accelerated", "mi355x", "internal", "fixtures")
    patch = os.path.join(fx, "groth16_rs_%s.patch" % curve)
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "FAILED" not in r.stdout, r.stdout + r.stderr
    src = open(patch).read()
    assert "TestingHooks.Randomness" in src and "TestingHooks.Solution(solution)" in src and "SetRandom" in src
    gen = open(os.path.join(fx, "gen_fixtures_test.go")).read()
    for name in ("meta.json", "pk.bin", "solution.bin", "r.bin", "s.bin", "proof.bin", "proof.raw"):   # the layout tests/gnark_fixture.py reads
        assert '"%s"' % name in gen, name
    import gnark_fixture
    assert all(k in gen for k in ("private_committed", "public_and_commitment_committed", "commitment_index", "nb_public", "curve"))
    assert gnark_fixture.run_case.__doc__


def test_go_shim_never_compares_jacobian_limbs():
    """ga_msm* return A Jacobian representative of the sum (include/gnark_amd.h): nothing in the Go shim may compare two Jacobian
    values with == / != (or use them as map keys): results go through FromJacobian / Equal.  Textual check of go/**: every identifier
    declared with a G1Jac / G2Jac type, and no comparison operator next to it."""
    import re
    bad = []
    for d, _, fs in os.walk(os.path.join(ROOT, "go")):
        for f in fs:
            if not f.endswith(".go"):
                continue
            src = open(os.path.join(d, f)).read()
            code = re.sub(r"//[^\n]*", "", src)
            names = set(re.findall(r"\bvar\s+(\w+)(?:\s*,\s*\w+)*\s+\w*\.?G[12]Jac\b", code)) | set(re.findall(r"\b(\w+)\s*:=\s*\w*\.?G[12]Jac\{", code))
            for m in re.finditer(r"\bvar\s+((?:\w+\s*,\s*)+\w+)\s+\w*\.?G[12]Jac\b", code):
                names |= set(x.strip() for x in m.group(1).split(","))
            for nm in names:
                if re.search(r"\b%s\s*(==|!=)|(==|!=)\s*&?%s\b" % (nm, nm), code):
                    bad.append((f, nm))
            assert not re.search(r"map\[\w*\.?G[12]Jac\]", code), f
    assert not bad, bad
    ga = open(os.path.join(ROOT, "go", "backend", "accelerated", "mi355x", "internal", "ga", "ga.go")).read()
    assert "never compare, hash or serialise the Jacobian limbs" in ga
