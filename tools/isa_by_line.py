#!/usr/bin/env python3
"""Static VALU / LDS / VMEM instruction counts of one kernel by SOURCE LINE (hipcc -gline-tables-only -save-temps assembly):
where a kernel's non-MFMA issue slots are spent.  `python tools/isa_by_line.py file.s <kernel-name-substring> [bucket]`."""
import re
import sys
from collections import Counter

path, want = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 1
s = open(path).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', s):
    files[int(m.group(1))] = m.group(3).split('/')[-1]
m = re.search(r'^(_Z\w*%s\w*):' % re.escape(want), s, flags=re.M)
i = m.start()
j = s.index('s_endpgm', i)
cur, c = None, {k: Counter() for k in ("valu", "lds", "vmem", "mfma")}
for l in s[i:j].split('\n'):
    mm = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if mm:
        cur = (files.get(int(mm.group(1)), '?'), int(mm.group(2)) // bucket * bucket)
        continue
    if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";")):
        op = l.split()[0]
        k = "mfma" if op.startswith('v_mfma') else "valu" if op.startswith('v_') else "lds" if op.startswith('ds_') else \
            "vmem" if op.startswith(('global_', 'buffer_', 'scratch_')) else None
        if k:
            c[k][cur] += 1
print(m.group(1)[:100])
print({k: sum(v.values()) for k, v in c.items()})
for (f, ln), n in sorted(c["valu"].items(), key=lambda x: -x[1])[:40]:
    print(f"{n:5d} valu  {c['lds'][(f, ln)]:4d} lds  {c['vmem'][(f, ln)]:4d} vmem  {c['mfma'][(f, ln)]:4d} mfma   {f}:{ln}")
