"""Lists the memory / wait / barrier / MFMA instructions of ONE kernel in program order, from the gfx950 assembly that
`hipcc -save-temps=obj` leaves next to the object — the quickest way to see loads that were meant to travel together but
compiled to load / s_waitcnt vmcnt(0) pairs (DESIGN.md §4.2 rule 19).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 ... -save-temps=obj -c easydgl_amd/csrc/k_score.hip -o /tmp/k_score.o
    python tools/isa_events.py /tmp/k_score-hip-amdgcn-amd-amdhsa-gfx950.s score_bwd_kernelIDF16bLi8ELi2 [max events]
Output: "<line offset in the kernel> <opcode> [wait counters]"."""
import re
import sys


def main():
    txt = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2]
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 80
    start = None
    for i, line in enumerate(txt):
        if line.startswith("_Z") and ":" in line and pat in line.split(":")[0]:
            start = i
            print(line.split(":")[0])
            break
    if start is None:
        sys.exit("no kernel label contains %r" % pat)
    shown = 0
    for i in range(start + 1, len(txt)):
        line = txt[i]
        if line.startswith(".Lfunc_end"):
            break
        if re.search(r"global_load|global_store|s_load|s_waitcnt|s_barrier|v_mfma|ds_write|ds_read|buffer_load|scratch_", line):
            parts = line.split(None, 1)
            arg = parts[1][:40] if len(parts) > 1 and "waitcnt" in parts[0] else ""
            print(i - start, parts[0], arg)
            shown += 1
            if shown >= limit:
                break


main()
