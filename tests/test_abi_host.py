"""CPU-side checks of the product's host code: the C-ABI library loads and exports every
symbol include/witch_hip.h declares; digitisation; no compute calls (no GPU here)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from witch_amd import _lib
    _lib.build()
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "witch_hip.h")).read()
    declared = set(re.findall(r"\b(wh_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.wh_version()


def test_digitize_matches_easel_rules():
    from witch_amd import _lib
    L = _lib.lib()
    out = np.zeros(13, np.uint8)
    assert L.wh_digitize(0, b"ACGTUNRYacgtX", 13, out.ctypes.data) == 0
    assert out.tolist() == [0, 1, 2, 3, 3, 15, 5, 6, 0, 1, 2, 3, 15]
    out = np.zeros(26, np.uint8)
    assert L.wh_digitize(2, b"ACDEFGHIKLMNPQRSTVWYBJZOUX", 26, out.ctypes.data) == 0
    assert out.tolist() == list(range(20)) + [21, 22, 23, 24, 25, 26]
    out = np.zeros(3, np.uint8)
    assert L.wh_digitize(0, b"A?C", 3, out.ctypes.data) == 1 and out[1] == 255


def test_product_does_not_use_the_oracle():
    """The product path must never route through oracle/ (that would void parity claims)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "witch_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "p7_oracle", "libp7oracle"):
                    assert needle not in text, (f, needle)


def test_only_tests_smoke_and_the_cpu_baseline_leg_import_the_oracle():
    """oracle/ is test infrastructure: outside tests/ only __graft_entry__.py (build + smoke) and bench.py (its cpu_baseline
    leg) may name it - no program under tools/, witch_amd/ or the repository root."""
    allowed = {os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")}
    for top in ("tools", "witch_amd", "."):
        base = os.path.join(ROOT, top)
        for dirpath, dirs, files in os.walk(base):
            if top == ".":
                dirs[:] = []           # the root's own files only
            for f in files:
                path = os.path.normpath(os.path.join(dirpath, f))
                if not f.endswith((".py", ".sh")) or path in allowed:
                    continue
                text = open(path).read()
                for needle in ("import oracle", "from oracle", "libp7oracle"):
                    assert needle not in text, (path, needle)


def test_no_gpu_fails_loudly_instead_of_falling_back():
    """On a box without a GPU the product must raise, never compute on the CPU."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tests.conftest import load_case
    from witch_amd.ehmm import EHMM
    from witch_amd._lib import WitchHipError
    case = load_case("dna_synth")
    with pytest.raises(WitchHipError) as ei:
        EHMM(case.hmm_paths)
    assert "no HIP device" in str(ei.value) or "wh_init" in str(ei.value)


def test_error_codes_and_messages_without_a_device():
    from witch_amd import _lib
    L = _lib.lib()
    assert L.wh_digitize(7, b"ACGT", 4, np.zeros(4, np.uint8).ctypes.data) < 0     # unknown alphabet
    assert b"bad argument" in L.wh_last_error()
    assert L.wh_ehmm_count(None) < 0
    assert L.wh_score(None, None, None, 0, None, None, None, None) < 0
    assert b"wh_score" in L.wh_last_error()


def test_sweep_context_is_register_passed(tmp_path):
    """The non-inlined sweeps of wh_score7.hip must receive their context struct in argument
    registers: when the aggregate stops flattening (a 17th dword, byte-sized members) the compiler
    passes it by reference through scratch and the callee faults on gfx950
    (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION).  Checked on the generated ISA."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "witch_amd", "csrc", "wh_score7.hip")
    out = tmp_path / "k7.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-DWH_SLIM_SPEC",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "witch_amd", "csrc"),
                        "--cuda-device-only", "-S", "-o", str(out), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    name, n_funcs, bad = None, 0, []
    for line in open(out):
        m = re.match(r"^(_ZN2wh2k7\d+(sweep_|region_scan)\S*):", line)
        if m:
            name = m.group(1)
            n_funcs += 1
        elif line.startswith("_Z") or "s_setpc_b64" in line:
            name = None if "s_setpc_b64" in line else name
        elif name and re.search(r"scratch_load_\w+ v\S*, v0, off", line):
            bad.append(name)
    assert n_funcs >= 20
    assert not bad, sorted(set(bad))[:3]
