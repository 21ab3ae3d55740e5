#!/bin/bash
# usage (GPU box): tools/round_profiles.sh r02  -> gpurun_out/profiles_<round>/ : rocprofv3 kernel stats + PMC summaries of
# c3 and c5, phase stamps (diagnostic build), shard / split sweeps, ablation timings.  Copy what should be judged into profiles/.
set -e
RND=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles_$RND; mkdir -p $OUT
# PARTS (default "prof rest"): "prof" = the rocprofv3 passes (about 4 minutes per config), "rest" = everything else
PARTS=${PARTS:-prof rest}
if [[ " $PARTS " == *" prof "* ]]; then
for c in ${CFGS:-c3 c5 c3sh}; do
  tools/prof.sh ${RND}_$c --config $c > $R/gpurun_out/prof_${RND}_$c.log 2>&1
  python tools/summarize_prof.py $R/gpurun_out/prof_${RND}_$c $OUT ${RND}_$c > /dev/null
  echo "profiled $c"
done
fi
[[ " $PARTS " == *" rest "* ]] || exit 0
hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -fPIC -shared -I$R/include -I$R/rt-gaussian-splat-renderer_amd/csrc -DRTGS_STAMPS=1 \
  $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_trace.hip $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_ops.hip -o $R/gpurun_out/lib_stamps.so 2>/dev/null
{
  echo "# k_trace per-phase s_memtime stamps (diagnostic -DRTGS_STAMPS=1 build, wave 0 of every workgroup), MI355X, round $RND"
  echo "# ticks = shader cycles (s_memtime; the clock line calibrates them against the 100 MHz s_memrealtime); never quote this build's run time"
  for c in c3 c5 c3sh; do echo "== $c"; RTGS_CFG=$c RTGS_HIP_LIB=$R/gpurun_out/lib_stamps.so python tools/stamps.py 2>&1 | grep -v amdgpu.ids; done
  echo "== one workgroup alone on the GPU (tools/one_tile.py), c3"
  RTGS_HIP_LIB=$R/gpurun_out/lib_stamps.so python tools/one_tile.py 2>&1 | grep tile
} > $OUT/${RND}_phase_stamps.txt
echo "stamps done"
{
  echo "# tools/shard_time.py (production build, HIP events, k_trace of rank 0's shard alone)"
  python tools/shard_time.py 2>&1 | grep shard
  echo "# tools/shard_loop.py: rank 0 of 8's frame loop on one GPU (real loop + real RCCL call for one peer's worth / no gather)"
  python tools/shard_loop.py c3 c5 2>&1 | grep "rank 0"
  echo "# tools/rank_loops.py: EVERY rank's loop on one GPU, equal and link-aware deals, N = 2 / 4 / 8; the frame times are PROJECTIONS (assumed link rates)"
  python tools/rank_loops.py c3 c5 2>&1 | grep "==\|N="
} > $OUT/${RND}_shards.txt
echo "shards done"
rm -f $R/gpurun_out/variants/results.txt
tools/variants.sh "full:" "ablate1_no_insertion:-DRTGS_ABLATE=1" "ablate2_no_hit_path:-DRTGS_ABLATE=2" "ablate3_no_candidate_loop:-DRTGS_ABLATE=3" "ablate4_scan_only:-DRTGS_ABLATE=4" > /dev/null 2>&1
cp $R/gpurun_out/variants/results.txt $OUT/${RND}_c3_ablation.txt
# (the summaries just made go where bench.py looks for them, so that the lines below carry roofline.frac)
for c in c3 c5 c3sh; do [ -f $OUT/${RND}_${c}_summary.json ] && cp $OUT/${RND}_${c}_summary.json $R/profiles/; done
for c in c1 c2 c3 c5 cap c3sh; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1; done > $OUT/${RND}_bench_lines_n1.jsonl
ls $OUT
