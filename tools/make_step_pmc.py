"""Three rocprofv3 --kernel-trace --pmc passes over the default bench command (tools/r06_call1.sh) -> the per-kernel table of
profiles/rNN_step_pmc.txt (the format tools/make_valu_floor.py reads).
   python tools/make_step_pmc.py PASS1.db PASS2.db PASS3.db [round tag] > profiles/rNN_step_pmc.txt
pass 1: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass 2: SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
pass 3: SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"""
import sqlite3
import sys
from collections import defaultdict


def means(path):
    db = sqlite3.connect(path)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for name, ctr, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
        a = acc[name][ctr]
        a[0] += val
        a[1] += 1
    return {k: {c: s / max(1, n) for c, (s, n) in v.items()} for k, v in acc.items()}


def main():
    p1, p2, p3 = (means(a) for a in sys.argv[1:4])
    tag = sys.argv[4] if len(sys.argv) > 4 else "r06"
    print(f"# Round {tag[1:].lstrip('0')}: where the wave cycles of the headline step's kernels go.  rocprofv3 --kernel-trace --pmc, three passes of 8 SQ counters")
    print("# (tools/r06_call1.sh), means per dispatch; SQ_WAVE_CYCLES / WAIT_* / ACTIVE_* count quad-cycles summed over a dispatch's waves.")
    print("# parked = SQ_WAIT_ANY (s_waitcnt / barrier), stall = SQ_WAIT_INST_ANY (issue stalls), issue = SQ_ACTIVE_INST_ANY — shares of SQ_WAVE_CYCLES;")
    print("# valu / lds = SQ_ACTIVE_INST_VALU / _LDS shares; per-wave instruction counts from SQ_INSTS_* / SQ_WAVES.")
    print("# kernel | waves | kcycles/wave | parked % | stall % | issue % | valu % | lds % | VALU insts/wave | LDS insts/wave | VMEM rd/wave | VMEM wr/wave | LDS bank conflict % ")
    rows = []
    for k, a in p1.items():
        b, c = p2.get(k, {}), p3.get(k, {})
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        waves = c.get("SQ_WAVES", 0.0)
        if wc <= 0 or waves <= 0:
            continue
        sh = lambda x: 100.0 * a.get(x, 0.0) / wc
        wc2 = b.get("SQ_WAVE_CYCLES", wc) or wc
        per = lambda x: b.get(x, 0.0) / waves
        rows.append((wc, f"{k[:96]} | {waves:.0f} | {4 * wc / waves / 1e3:.1f} | {sh('SQ_WAIT_ANY'):.1f} | {sh('SQ_WAIT_INST_ANY'):.1f} | "
                         f"{sh('SQ_ACTIVE_INST_ANY'):.1f} | {sh('SQ_ACTIVE_INST_VALU'):.1f} | {sh('SQ_ACTIVE_INST_LDS'):.1f} | {per('SQ_INSTS_VALU'):.0f} | "
                         f"{per('SQ_INSTS_LDS'):.0f} | {per('SQ_INSTS_VMEM_RD'):.0f} | {per('SQ_INSTS_VMEM_WR'):.0f} | "
                         f"{100.0 * b.get('SQ_LDS_BANK_CONFLICT', 0.0) / wc2:.1f}"))
    for _, line in sorted(rows, key=lambda r: -r[0]):
        print(line)


if __name__ == "__main__":
    main()
