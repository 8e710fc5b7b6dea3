#!/usr/bin/env python3
"""Per-kernel instruction statistics of a hipcc `-save-temps` assembly file (*.s): total instructions, MFMA / VALU / LDS /
VMEM / SALU counts, registers, LDS bytes, spills.  `python tools/isa_stats.py file.s [substring-of-kernel-name]`."""
import re
import sys
from collections import Counter


def kernels(path):
    cur, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".") and not line.startswith("\t"):
            if cur and body:
                yield cur, body
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(line)
    if cur and body:
        yield cur, body


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "sync"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    for name, body in kernels(path):
        if want and want not in name:
            continue
        ops = [l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";")) and l.split()]
        if not any(o.startswith("s_endpgm") for o in ops):
            continue
        c = Counter(classify(o) for o in ops)
        meta = {}
        for l in body:
            for k in ("NumVgprs", "NumAgprs", "NumSgprs", "ScratchSize", "LDSByteSize", "Occupancy"):
                m = re.search(rf"; {k}: (\d+)", l)
                if m:
                    meta[k] = int(m.group(1))
        top = Counter(o for o in ops if classify(o) == "valu").most_common(8)
        print(f"{name[:110]}\n   total {len(ops)}  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())) + f"\n   {meta}\n   top VALU: {top}")


if __name__ == "__main__":
    main()
