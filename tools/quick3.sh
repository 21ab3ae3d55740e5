#!/bin/bash
# usage (GPU box): tools/quick3.sh <outdir> [configs...]  -- frame / k_trace / project+bin ms of the headline configs
OUT=$1; shift; mkdir -p $OUT
for c in ${@:-c3 c5 c3sh}; do
  python bench.py --config $c --no-cpu-baseline --no-secondary --no-frame-batch 2>/dev/null | tail -1 > $OUT/line_$c.json
  python -c "import json;d=json.load(open('$OUT/line_$c.json'));print('$c frame %.4f  k_trace %.4f  project+bin %.4f ms  (%.0f Mrays/s)'%(d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['project_ms'],d['value']))" | tee -a $OUT/times.txt
done
