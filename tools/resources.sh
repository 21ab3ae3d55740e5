#!/bin/bash
# usage: tools/resources.sh [extra hipcc flags]  -- register / spill / scratch / occupancy table of every kernel in
# rtgs_trace.hip (clang -Rpass-analysis=kernel-resource-usage), cross-compiled for gfx950: no GPU needed
R=$(cd "$(dirname "$0")/.." && pwd)
hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -fPIC -c -I$R/include -I$R/rt-gaussian-splat-renderer_amd/csrc "$@" \
  -Rpass-analysis=kernel-resource-usage $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_trace.hip -o /dev/null 2>&1 |
python3 -c '
import re, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: (.*)", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        k = re.search(r"(k_[a-z_]+)(IL[^E]*E)?", name)
        short = name
        if k:
            short = k.group(1)
            tp = re.findall(r"L[bi](\d+)E", name)
            if tp: short += "<" + ",".join(tp) + ">"
        cur = {"name": short}; rows.append(cur)
    elif cur is not None and ":" in t:
        a, b = t.split(":", 1); cur[a.strip()] = b.replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
cols = ["VGPRs", "TotalSGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
print("%-28s " % "kernel" + " ".join("%10s" % c.split(" [")[0][:10] for c in cols))
for r in rows:
    print("%-28s " % r["name"] + " ".join("%10s" % r.get(c, "?") for c in cols))
'
