#!/bin/bash
# rocprofv3 kernel trace of the default bench (all three N = 1 queueing modes): per (kernel, grid) durations.
# usage: tools/prof_batched.sh <outdir> [config]
OUT=$1; CFG=${2:-c3}; mkdir -p $GRAFT_REPO_ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --no-cpu-baseline --no-secondary --min-timed-s 0.3 > $GRAFT_REPO_ROOT/$OUT/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/trace_groups.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1)
