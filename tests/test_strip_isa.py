"""The strip scoring kernels (csrc/k_score_strip.hip, csrc/k_score_stripw.hip) issue their MFMAs from inline asm, so hipcc neither sees nor pads the hazards
around them (guide §5.7): this test rebuilds the assembly and audits it with tools/isa_hazards.py — no VALU / accvgpr write of an
MFMA source within 3 instructions before it, no write of a SrcC != D or non-MFMA read of a result within 16 wait states after it —
and checks that the hot loops carry no scratch (spill) traffic and no compiler v_accvgpr moves.  CPU only (hipcc cross-compiles)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def strip_asm():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(ROOT, "easydgl_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("_edgl_build", os.path.join(ROOT, "easydgl_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    d = tempfile.mkdtemp(prefix="strip_isa_")
    flags = b.FLAGS + b.EXTRA_FLAGS["k_score_strip.hip"]
    src = os.path.join(ROOT, "easydgl_amd", "csrc", "k_score_strip.hip")
    r = subprocess.run([HIPCC] + flags + ["-save-temps=obj", "-c", src, "-o", os.path.join(d, "strip.o")], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr[-3000:]
    path = os.path.join(d, "k_score_strip-hip-amdgcn-amd-amdhsa-gfx950.s")
    assert os.path.exists(path)
    yield path
    shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("kernel", ["strip_kernelILi0", "strip_kernelILi1", "fallback_exact"])
def test_no_unpadded_mfma_hazards(strip_asm, kernel):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazards.py"), strip_asm, kernel, "3", "16"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    last = r.stdout.strip().splitlines()[-1]
    m = re.match(r"(\d+) MFMAs checked, (\d+) findings", last)
    assert m, r.stdout[-2000:]
    assert int(m.group(1)) >= 90 and int(m.group(2)) == 0, r.stdout[-3000:]


@pytest.mark.parametrize("kernel", ["strip_kernelILi0", "strip_kernelILi1"])
def test_hot_loops_are_free_of_spills_and_accumulator_moves(strip_asm, kernel):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_loops.py"), strip_asm, kernel, "100"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split() for l in r.stdout.splitlines()[2:] if l.startswith(".LBB") or l.startswith("entry")]
    hot = [w for w in rows if int(w[2]) >= 13]          # blocks that issue MFMAs of the main loop (>= 13 per block)
    assert hot, r.stdout
    mfma = sum(int(w[2]) for w in hot)
    assert mfma >= 128
    for w in hot:
        # columns: block n mfma valu trans acc ldsr ldsw vmem scratch ...
        assert int(w[9]) == 0, ("scratch traffic in a hot block", w)
        assert int(w[5]) <= 128, ("accumulator moves in a hot block", w)      # (the prologue block zero-fills 128 AGPRs)
    loop_blocks = [w for w in hot if "back" in " ".join(w) or int(w[2]) >= 32]
    assert all(int(w[5]) == 0 for w in loop_blocks), loop_blocks


# ---- the wide form (C = 256): k_score_stripw.hip — LDS-direct staging, so the hot loop must also be free of ds_write and of
# ---- compiler-visible global loads (every vmcnt in it is placed by hand)
@pytest.fixture(scope="module")
def stripw_asm():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location("_edgl_build_w", os.path.join(ROOT, "easydgl_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    d = tempfile.mkdtemp(prefix="stripw_isa_")
    flags = b.FLAGS + b.EXTRA_FLAGS["k_score_stripw.hip"]
    src = os.path.join(ROOT, "easydgl_amd", "csrc", "k_score_stripw.hip")
    r = subprocess.run([HIPCC] + flags + ["-save-temps=obj", "-c", src, "-o", os.path.join(d, "stripw.o")], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr[-3000:]
    path = os.path.join(d, "k_score_stripw-hip-amdgcn-amd-amdhsa-gfx950.s")
    assert os.path.exists(path)
    yield path
    shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("kernel", ["stripw_kernelILi0", "stripw_kernelILi1", "fallback_exactILi256", "stripw5_kernelILi0", "stripw5_kernelILi1", "fallback_exact5"])
def test_wide_kernels_have_no_unpadded_mfma_hazards(stripw_asm, kernel):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazards.py"), stripw_asm, kernel, "3", "16"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    last = r.stdout.strip().splitlines()[-1]
    m = re.match(r"(\d+) MFMAs checked, (\d+) findings", last)
    assert m, r.stdout[-2000:]
    assert int(m.group(1)) >= 90 and int(m.group(2)) == 0, r.stdout[-3000:]


@pytest.mark.parametrize("kernel", ["stripw_kernelILi0", "stripw_kernelILi1", "stripw5_kernelILi0", "stripw5_kernelILi1"])
def test_wide_hot_loops_are_free_of_spills_stores_and_accumulator_moves(stripw_asm, kernel):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_loops.py"), stripw_asm, kernel, "100"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split() for l in r.stdout.splitlines()[2:] if l.startswith(".LBB") or l.startswith("entry")]
    # columns: block n mfma valu trans acc ldsr ldsw vmem scratch ...
    # the two iterations of a trip: >= 20 MFMAs and >= 40 LDS reads each (the prologue block, with the x-fragment loads, is not one)
    loop = [w for w in rows if int(w[2]) >= 20 and int(w[6]) >= 40 and int(w[8]) <= 12]
    assert len(loop) >= 2, r.stdout
    assert sum(int(w[2]) for w in loop) >= 56
    for w in loop:
        assert int(w[9]) == 0, ("scratch traffic in the loop", w)
        assert int(w[5]) == 0, ("accumulator moves in the loop", w)
        assert int(w[7]) == 0, ("ds_write in the loop: the unit is staged by global_load_lds", w)
        assert int(w[8]) == (9 if "stripw5" in kernel else 5), ("a trip half issues its 4 (C = 512: 8) block loads + the C operands, nothing else", w)
