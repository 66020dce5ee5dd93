#!/usr/bin/env python3
"""min .. max per setting of gpurun_out/sustained_ab.txt (tools/sustained_ab.sh)"""
import re
import sys

cur, res = None, {}
for line in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sustained_ab.txt"):
    if line.startswith("=="):
        cur = line.split(": ", 1)[1].strip()
    m = re.search(r"([\d.]+) us per step, match launch\s+([\d.]+)", line)
    if m:
        res.setdefault(cur, []).append((float(m.group(1)), float(m.group(2))))
for k, v in res.items():
    print(f"{k:55s} step {min(a for a, b in v):7.2f} .. {max(a for a, b in v):7.2f} us   "
          f"match {min(b for a, b in v):6.2f} .. {max(b for a, b in v):6.2f}")
