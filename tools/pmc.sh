#!/bin/bash
# usage: tools/pmc.sh <tag> <lib.so|-> "<counters pass 1>" "<counters pass 2>" ...
set -e
TAG=$1; LIB=$2; shift 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
[ "$LIB" != "-" ] && export RTGS_HIP_LIB=$R/$LIB
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-frame-batch --min-timed-s 0.1 ${BENCH_ARGS} > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:28s} {sum(v)/len(v):16.0f}")
PY
