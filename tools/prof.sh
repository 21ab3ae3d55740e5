#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]; kernel-trace stats + two PMC passes -> gpurun_out/prof_<tag>
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-frame-batch --min-timed-s 0.1 "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-frame-batch --min-timed-s 0.1 "$@" > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-frame-batch --min-timed-s 0.1 "$@" > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-frame-batch --min-timed-s 0.1 "$@" > $OUT/bench_pmc_write.log 2>&1
# vector-memory side (round 4): TA / TCP (vL1D) activity, what binds the SH instantiations (c5, c3sh, every PLY capture).
# Two counters of a block per pass (five TA counters in one: "exceeds the capabilities of the hardware", and rocprofv3
# then hangs in its abort handler: every pass runs under a timeout).
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-frame-batch --min-timed-s 0.1"
i=0
for pass in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_v$i -- $B "$@" > $OUT/bench_pmc_v$i.log 2>&1 || echo "pass $i ($pass) failed"
  echo "pass $i done" >> $OUT/progress.log
done
find $OUT -name "*.csv" | head -40
