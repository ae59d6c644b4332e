#!/usr/bin/env python3
"""Register / scratch / occupancy figures the compiler reports per kernel of one .hip file (no GPU needed).
usage: scripts/kernel_resources.py path-tracer_amd/csrc/trace_kernels.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC"] + extra + \
      ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
name, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        rows[name] = {}
        continue
    m = re.search(r"remark: +(VGPRs|AGPRs|TotalSGPRs|ScratchSize|Occupancy|LDS Size)[^:]*: (\d+)", line)
    if m and name and "Spill" not in line:
        rows[name][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    print("%-52s vgpr %4d sgpr %4d scratch %4d occ %d lds %d" % (k, v.get("VGPRs", 0), v.get("TotalSGPRs", 0), v.get("ScratchSize", 0),
                                                               v.get("Occupancy", 0), v.get("LDS Size", 0)))
if not rows:
    sys.stderr.write(out)
