#!/bin/bash
# One GPU call: the whole -m gpu suite, then (unless the suite was killed / aborted) bench lines.
#   usage: tools/run_r5.sh <tag> [pytest -k expr]      env: BENCH="c3 c5 ..." (default: c3 default run), STEPS
TAG=${1:?tag}; KEXPR=${2:-}
OUT=gpurun_out/$TAG; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.txt 2>&1 || { tail -5 $OUT/build.txt; exit 1; }
if [ -n "$KEXPR" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/tests.txt 2>&1; rc=$?
else
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1; rc=$?
fi
tail -15 $OUT/tests.txt
echo "tests rc $rc"
if [ $rc -ge 124 ]; then exit $rc; fi        # timed out / killed / aborted: nothing further on this box
for cfg in ${BENCH:-c3}; do
  timeout -k 10 400 python bench.py --config $cfg ${STEPS:+--steps $STEPS} > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; brc=$?
  echo "bench $cfg rc $brc"; tail -c 600 $OUT/bench_$cfg.err
  python - <<PY
import json
try:
    l = json.loads(open("$OUT/bench_$cfg.json").read().strip().splitlines()[-1])
    r = l["roofline"]
    print("$cfg", "Mrays/s %.0f" % l["value"], "ms %.4f" % l["ms_per_step"], "timed_s %.2f" % l["timed_s"], "k_trace %.4f" % r["kernel_ms"],
          "project %.4f" % r["project_ms"], "frac", r["frac"], r["bound"], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in l["secondary"].items() if "ms" in k})
    print("cpu", {k: v for k, v in l.get("cpu_baseline", {}).items() if k in ("value", "c_openmp")})
except Exception as e:
    print("no line:", e)
PY
  if [ $brc -ge 124 ]; then exit $brc; fi
done
