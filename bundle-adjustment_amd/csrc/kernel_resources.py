#!/usr/bin/env python3
"""hipcc -Rpass-analysis=kernel-resource-usage remarks (one .res file per translation unit) -> JSON on stdout:
{kernel symbol (demangled where c++filt is there): {vgprs, agprs, sgprs, vgpr_spills, sgpr_spills, scratch_bytes_per_lane, lds_bytes, occupancy}}.
Written by the Makefile next to libjaicov_neq.so; bench.py quotes the kernels of the LM pass from it."""
import json
import re
import subprocess
import sys

KEYS = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "VGPRs Spill": "vgpr_spills", "SGPRs Spill": "sgpr_spills",
        "ScratchSize [bytes/lane]": "scratch_bytes_per_lane", "LDS Size [bytes/block]": "lds_bytes", "Occupancy [waves/SIMD]": "occupancy"}
out = {}
cur = None
for path in sys.argv[1:]:
    try:
        lines = open(path, errors="replace").read().splitlines()
    except OSError:
        continue
    for ln in lines:
        m = re.search(r"remark: (?:\s*)Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {"unit": path.rsplit(".", 1)[0]})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", ln)
        if m and cur is not None and m.group(1).strip() in KEYS:
            cur[KEYS[m.group(1).strip()]] = int(m.group(2))
names = list(out)
try:
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    out = {d: out[n] for n, d in zip(names, dem)}
except Exception:
    pass
json.dump(out, sys.stdout, indent=1, sort_keys=True)
