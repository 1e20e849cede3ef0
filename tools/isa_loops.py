"""Per-basic-block instruction mix of ONE kernel from the gfx950 assembly `hipcc -save-temps=obj` leaves behind:
    python tools/isa_loops.py /tmp/<file>-hip-amdgcn-amd-amdhsa-gfx950.s <kernel name substring> [min instructions]
For every block (label to label) with at least `min` instructions: MFMA / VALU / transcendental / accvgpr moves / LDS reads and
writes / global / scratch / s_waitcnt / s_nop / SALU counts and whether the block ends in a backward branch (a loop body).
Used to check a hand-placed stream: issue slots per MFMA, spills inside the loop, AGPR traffic."""
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "acc"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", op):
        return "trans"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ldsr"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "ldsw"
    if op.startswith("ds_"):
        return "ldsx"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op == "s_waitcnt":
        return "wait"
    if op == "s_nop":
        return "nop"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    txt = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2]
    minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    start = None
    for i, line in enumerate(txt):
        if line.startswith("_Z") and ":" in line and pat in line.split(":")[0]:
            start = i
            print(line.split(":")[0])
            break
    if start is None:
        sys.exit("no kernel label contains %r" % pat)
    blocks, cur, name, order = [], {}, "entry", {}
    n = 0
    for i in range(start + 1, len(txt)):
        line = txt[i].strip()
        if line.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            blocks.append((name, cur, n))
            name, cur, n = m.group(1), {}, 0
            order[name] = len(blocks)
            continue
        if not line or line.startswith(";") or line.startswith("."):
            continue
        op = line.split()[0]
        k = classify(op)
        cur[k] = cur.get(k, 0) + 1
        n += 1
        if k == "branch":
            tgt = line.split()[-1]
            cur.setdefault("_targets", []).append(tgt)
    blocks.append((name, cur, n))
    keys = ["mfma", "valu", "trans", "acc", "ldsr", "ldsw", "vmem", "scratch", "wait", "nop", "salu", "barrier", "branch"]
    print("%-12s %5s " % ("block", "n") + " ".join("%7s" % k for k in keys) + "  loop")
    for idx, (name, c, n) in enumerate(blocks):
        if n < minlen:
            continue
        back = any(order.get(t, 1 << 30) <= idx for t in c.get("_targets", []))
        print("%-12s %5d " % (name, n) + " ".join("%7d" % c.get(k, 0) for k in keys) + ("  <-- back edge" if back else ""))


main()
