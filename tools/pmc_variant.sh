#!/bin/bash
# PMC passes of one bench config with a BUILD VARIANT of the library (built on the box).
# usage: tools/pmc_variant.sh <tag> "<hipcc flags>" <config> "<counters pass 1>" ...
TAG=$1; FLAGS=$2; CFG=$3; shift 3
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmcv_$TAG; mkdir -p $OUT
hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -fPIC -shared -I$R/include -I$R/rt-gaussian-splat-renderer_amd/csrc $FLAGS \
  $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_trace.hip $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_ops.hip -o $OUT/lib.so 2>/dev/null || { echo "BUILD FAILED"; exit 1; }
export RTGS_HIP_LIB=$OUT/lib.so
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- python3 $R/bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-frame-batch --min-timed-s 0.1 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("variant $TAG ($FLAGS), config $CFG: k_trace PMC means per launch")
for k, v in sorted(agg.items()):
    print(f"  {k:28s} {sum(v)/len(v):16.0f}")
PY
