"""Builds libeasydgl_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain
C-ABI shared object (include/easydgl_hip.h)."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libeasydgl_hip.so")
SOURCES = ["k_misc.hip", "k_data.hip", "k_encode.hip", "k_gemm.hip", "k_gemm2.hip", "k_layernorm.hip", "k_bimau_fwd.hip", "k_bimau_bwd.hip", "k_bimau_big.hip",
           "k_score.hip", "k_score_strip.hip", "k_score_stripw.hip", "k_eval_topk.hip", "k_tattn.hip", "k_coding.hip", "k_tail.hip"]
HEADERS = ["edgl_common.h", "batch_prep.h", "score_plan.h", "topk_select.h", "gemm_tile.h", "bimau_common.h", "bimau_fwd_impl.h", "bimau_bwd_impl.h", os.path.join("..", "..", "include", "easydgl_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (no v_accvgpr_read/write traffic around every VALU consumer);
# gfx950's register file is unified, so nothing is lost by not using AGPRs.  Per file: hipcc 7.2 crashes on k_score.hip
# with it, and the GEMM kernels (pure accumulate-then-store) gain nothing.
EXTRA_FLAGS = {"k_bimau_fwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "k_bimau_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "k_bimau_big.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "k_tattn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               # hand-placed instruction stream: the SLP vectoriser packs the row sums into v_pk_add_f32 (slow beside MFMAs)
               "k_score_strip.hip": ["-fno-slp-vectorize", "-Wno-unused-value"] + (["-DSTRIP_SAFE"] if os.environ.get("EDGL_STRIP_SAFE") else []) + os.environ.get("EDGL_STRIP_DEFS", "").split(),
               "k_score_stripw.hip": ["-fno-slp-vectorize", "-Wno-unused-value"] + (["-DSTRIP_SAFE"] if os.environ.get("EDGL_STRIP_SAFE") else []) + os.environ.get("EDGL_STRIPW_DEFS", "").split()}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths, extra=()) -> str:
    """sha256 over the CONTENT of the inputs (sources, headers, flags): a prebuilt object with a fresh mtime but other
    sources is rebuilt, an untouched tree is not."""
    h = hashlib.sha256()
    for e in extra:
        h.update(str(e).encode() + b"\0")
    for p in paths:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def _stamp_ok(target: str, digest: str) -> bool:
    stamp = target + ".sha256"
    return os.path.exists(target) and os.path.exists(stamp) and open(stamp).read().strip() == digest


def _write_stamp(target: str, digest: str) -> None:
    with open(target + ".sha256", "w") as f:
        f.write(digest + "\n")


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    digests = {}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        digests[o] = _digest([s] + hdrs, FLAGS + EXTRA_FLAGS.get(src, []))
        if force or not _stamp_ok(o, digests[o]):
            jobs.append((s, o))

    def run(job):
        s, o = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr[-4000:]))
        _write_stamp(o, digests[o])
        if verbose:
            print("compiled", os.path.basename(s))
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    link_digest = _digest([], [digests[o] for o in objs])   # the library is current iff it was linked from THESE objects
    if force or jobs or not _stamp_ok(OUT, link_digest):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        _write_stamp(OUT, link_digest)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
