#!/bin/bash
# phase stamps (diagnostic -DRTGS_STAMPS=1 builds) of k_trace variants: usage tools/ab_stamps.sh "NAME:-DFLAG.." ...   env CFG=c3
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab; mkdir -p $OUT
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -fPIC -shared -I$R/include -I$R/rt-gaussian-splat-renderer_amd/csrc -DRTGS_STAMPS=1 $flags \
     $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_trace.hip $R/rt-gaussian-splat-renderer_amd/csrc/rtgs_ops.hip -o $OUT/libst_$name.so 2>/dev/null || { echo "$name: BUILD FAILED"; continue; }
  echo "== $name"
  RTGS_CFG=${CFG:-c3} RTGS_HIP_LIB=$OUT/libst_$name.so python $R/tools/stamps.py 2>&1 | grep -v amdgpu.ids | head -16
done
