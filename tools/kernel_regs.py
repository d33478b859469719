#!/usr/bin/env python3
"""Register / spill / LDS table of the kernels in a hipcc -save-temps device assembly file (developer tool).
    python tools/kernel_regs.py /tmp/wc_se-hip-amdgcn-amd-amdhsa-gfx950.s [name filter]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
    name, body = m.group(1), m.group(2)
    get = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, body) or [0, "?"])[1]
    rows.append((name, get("vgpr_count"), get("agpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"),
                 get("group_segment_fixed_size"), get("max_flat_workgroup_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (n, *rest), d in zip(rows, names):
    if flt in d:
        print("%-110s vgpr %s agpr %s spill %s sspill %s lds %s wg %s" % ((d[:110],) + tuple(rest)))
