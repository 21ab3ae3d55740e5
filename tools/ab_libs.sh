#!/bin/bash
# A/B of prebuilt libraries (tools/_ab/lib_<name>.so, built here with hipcc from any tree: tools/_ab/ is git-ignored but
# travels to the GPU box), alternating inside one call.  usage: tools/ab_libs.sh "name1 name2 ..."   env: CFGS="c3 c2", REPS=2
R=$GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-2}); do
 for name in $1; do
  for c in ${CFGS:-c3 c2}; do
    RTGS_HIP_LIB=$R/tools/_ab/lib_$name.so python $R/bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | \
     python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name $c rep$rep', 'ms %.4f trace_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
 done
done
