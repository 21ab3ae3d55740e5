#!/bin/bash
# rocprofv3 kernel trace of ONE rank's tile-parallel loop on one GPU (exchange stubbed out): per-kernel durations.
# usage: tools/prof_rank_loop.sh <outdir>   env: CFG WORLDS RANKS F B (as tools/rank_loops.py)
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DEALS=equal rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o loop -- python3 $GRAFT_REPO_ROOT/tools/rank_loops.py ${CFG:-c3} > $GRAFT_REPO_ROOT/$OUT/loop.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:12]:
    print("%-60s calls %6s  avg %9.1f ns  total %10.0f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
PY
grep "N=" $OUT/loop.txt | cut -c1-200
