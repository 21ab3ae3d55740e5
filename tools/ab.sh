#!/bin/bash
# A/B of k_trace build variants on the GPU box: c3 and c5 frame / kernel times (+ optional parity slice).
# usage: tools/ab.sh "NAME:-DFLAG=.. -DFLAG=.." ...     env: CFGS="c3 c5" (default), CHECK=1 (golden parity slice per variant)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab; mkdir -p $OUT
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -fPIC -shared -I$R/include -I$R/rt-gaussian-splat-renderer_amd/csrc $flags \
     $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_trace.hip $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_ops.hip -o $OUT/lib_$name.so 2>/dev/null || { echo "$name: BUILD FAILED"; continue; }
  if [ -n "$CHECK" ]; then
    RTGS_HIP_LIB=$OUT/lib_$name.so python -m pytest $R/tests -x -q -m gpu -k "golden or stacked or truncation or duplicates or ties or sharded or random or fuzz" > $OUT/check_$name.log 2>&1 \
      && echo "$name parity slice: $(tail -1 $OUT/check_$name.log)" || { echo "$name PARITY FAILED: $(tail -3 $OUT/check_$name.log)"; }
  fi
  for c in ${CFGS:-c3 c5}; do
    RTGS_HIP_LIB=$OUT/lib_$name.so python $R/bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-frame-batch 2>/dev/null | tail -1 | \
     python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name $c', 'ms %.4f trace_ms %.4f project_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['project_ms']))" | tee -a $OUT/results.txt
  done
done
