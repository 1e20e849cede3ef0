"""Audit of the hand-placed MFMA stream of a kernel (assembly from `hipcc -save-temps=obj`): hipcc neither sees nor pads an MFMA
written as inline asm (guide §5.7), so the wait states around each one are the author's.  For every v_mfma inside ;;#ASMSTART:
  RAW-in : a VALU / v_accvgpr write of one of its source registers within the previous `near` instructions   (needs s_nop 1+)
  WAR-C  : a write (VALU, LDS / VMEM load destination) of its SrcC registers within the next `far` instructions, when SrcC != D
  RAW-out: a non-MFMA read of its D registers within the next `far` instructions (the next MFMA taking D whole as SrcC is fine)
    python tools/isa_hazards.py <file.s> <kernel substring> [near=3] [far=14]"""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([va])(\d+)$", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


def parse(line):
    parts = line.split(None, 1)
    op = parts[0]
    ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
    return op, ops


def main():
    txt = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2]
    near = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    far = int(sys.argv[4]) if len(sys.argv) > 4 else 14
    start = next((i for i, l in enumerate(txt) if l.startswith("_Z") and ":" in l and pat in l.split(":")[0]), None)
    if start is None:
        sys.exit("kernel not found")
    ins = []
    for i in range(start + 1, len(txt)):
        l = txt[i].strip()
        if l.startswith(".Lfunc_end"):
            break
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            continue
        ins.append((i + 1, l))
    n_mfma = 0
    bad = 0
    for k, (ln, l) in enumerate(ins):
        op, ops = parse(l)
        if not op.startswith("v_mfma"):
            continue
        n_mfma += 1
        d, a, b, c = regs(ops[0]), regs(ops[1]), regs(ops[2]), regs(ops[3])
        src = a | b | c
        first = max(0, k - near)
        for j in range(k - 1, first - 1, -1):      # control does not fall through an unconditional branch: what stands in front of
            if parse(ins[j][1])[0] in ("s_branch", "s_endpgm", "s_setpc_b64"):      # it belongs to another path
                first = j + 1
                break
        for j in range(first, k):
            op2, ops2 = parse(ins[j][1])
            if op2.startswith("v_mfma") or op2.startswith("s_") or op2.startswith("ds_read") or op2.startswith("global_load") or not ops2:
                continue
            if op2.startswith("v_") and regs(ops2[0]) & src:
                print(f"RAW-in  line {ln}: {l}\n          <- {ins[j][0]}: {ins[j][1]}"); bad += 1
        states = 0
        for j in range(k + 1, len(ins)):
            op2, ops2 = parse(ins[j][1])
            states += (int(ops2[0]) + 1) if op2 == "s_nop" and ops2 else 1
            if states > far:
                break
            if not ops2 or op2.startswith("s_"):
                continue
            wr = regs(ops2[0]) if not (op2.startswith("ds_write") or op2.startswith("global_store")) else set()
            if c != d and wr & c and not op2.startswith("v_mfma"):
                print(f"WAR-C   line {ln}: {l}\n          -> {ins[j][0]}: {ins[j][1]}"); bad += 1
            if op2.startswith("v_mfma"):
                if regs(ops2[3]) == d:
                    continue
                rd = regs(ops2[1]) | regs(ops2[2]) | regs(ops2[3])
            else:
                rd = set().union(*[regs(t) for t in (ops2[1:] if wr else ops2)])
            if rd & d:
                print(f"RAW-out line {ln}: {l}\n          -> {ins[j][0]}: {ins[j][1]}"); bad += 1
    print(f"{n_mfma} MFMAs checked, {bad} findings")


main()
