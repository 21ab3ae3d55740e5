#!/bin/bash
# like ab.sh, but a variant whose parity slice fails (or faults) is not timed, and nothing further runs after it
# usage: tools/ab_safe.sh "NAME:flags" ...   env: CFGS="c3 c3sh c5"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab; mkdir -p $OUT
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -fPIC -shared -I$R/include -I$R/rt-gaussian-splat-renderer_amd/csrc $flags \
     $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_trace.hip $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_ops.hip -o $OUT/lib_$name.so 2>/dev/null || { echo "$name: BUILD FAILED"; exit 1; }
  RTGS_HIP_LIB=$OUT/lib_$name.so timeout -k 10 600 python -m pytest $R/tests/test_gpu_render.py -x -q -m gpu > $OUT/check_$name.log 2>&1 \
    || { echo "$name PARITY FAILED: $(grep -v '^Extension' $OUT/check_$name.log | tail -5)"; exit 1; }
  echo "$name render tests: $(tail -1 $OUT/check_$name.log)"
  for c in ${CFGS:-c3 c3sh c5}; do
    RTGS_HIP_LIB=$OUT/lib_$name.so python $R/bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | \
     python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name $c', 'ms %.4f trace_ms %.4f project_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['project_ms']))" | tee -a $OUT/results.txt || exit 1
  done
done
