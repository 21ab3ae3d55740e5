#!/bin/bash
# Build k_trace variants (compile-time switches) and bench each on the GPU box.
# usage: tools/variants.sh "NAME:-DFLAG=.. -DFLAG=.." ...
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/variants; mkdir -p $OUT
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -fPIC -shared -I$R/include -I$R/rt-gaussian-splat-renderer_amd/csrc $flags \
     $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_trace.hip $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_ops.hip -o $OUT/lib_$name.so
  RTGS_HIP_LIB=$OUT/lib_$name.so python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary ${BENCH_ARGS} 2>/dev/null | tail -1 | \
     python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'Mrays/s %.0f ms %.4f trace_ms %.4f project_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['project_ms']))" | tee -a $OUT/results.txt
  if [ -n "$SHARD" ]; then RTGS_HIP_LIB=$OUT/lib_$name.so python $R/tools/shard_time.py 2>&1 | grep shard | tee -a $OUT/results.txt; fi
done
