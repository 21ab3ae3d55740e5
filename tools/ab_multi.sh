#!/bin/bash
# A/B of the batched-frames form (kMulti) by occupancy: every rank's loop at N = 8 for each build variant.
# usage: tools/ab_multi.sh <outdir> "NAME:-DFLAG=.." ...    env: CFG (c3), WORLDS (8), F ("2"), B ("8")
R=$GRAFT_REPO_ROOT; OUT=$1; shift; mkdir -p $OUT
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -fPIC -shared -I$R/include -I$R/rt-gaussian-splat-renderer_amd/csrc $flags \
     $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_trace.hip $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_ops.hip -o $OUT/lib_$name.so 2>/dev/null || { echo "$name: BUILD FAILED"; continue; }
  echo "== $name ($flags)"
  RTGS_HIP_LIB=$OUT/lib_$name.so WORLDS="${WORLDS:-8}" F="${F:-2}" B="${B:-8}" timeout -k 10 300 python $R/tools/rank_loops.py ${CFG:-c3} 2>&1 | grep "N=\|==" | cut -c1-250
done
