#!/bin/bash
# usage (on the GPU box): tools/quick.sh <tag> [pytest -k expr]  -> parity slice + c3/c5 bench lines under gpurun_out/
TAG=$1; K=${2:-"golden or analytic or stacked or truncation or duplicates or two_waves or sharded or c3 or fuzz"}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
python -m pytest $R/tests -x -q -m gpu -k "$K" > $R/gpurun_out/q_${TAG}_tests.log 2>&1; rc=$?
tail -4 $R/gpurun_out/q_${TAG}_tests.log
[ $rc -ne 0 ] && exit $rc
for c in c3 c5 c2; do
  python $R/bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $R/gpurun_out/q_${TAG}_$c.json
  python - <<PY
import json
d=json.loads(open("$R/gpurun_out/q_${TAG}_$c.json").read())
print("$TAG $c Mrays/s %.0f ms %.4f trace_ms %.4f project_ms %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["project_ms"]))
PY
done
