#!/bin/bash
# A/B of kernel variants on one GPU box (scratch helper; results land in gpurun_out/)
mkdir -p gpurun_out
run() { echo "== $*"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  step %.3f ms  fwd %.3f ms (%.1f%%)  bwd %.3f ms (%.1f%%)' % (d['ms_per_step'], r['avg_launch_ms'], 100*r['frac'], r['backward']['avg_ms'], 100*r['backward']['frac']))
"; }
run
run --tuning roi_align_fwd_slices=4
run --tuning roi_align_fwd_slices=2
run --tuning roi_align_fwd_slices=1
run --tuning roi_align_fwd_slices=16
run --tuning roi_align_fwd=0
run --tuning roi_align_bwd=0
run --tuning roi_align_bwd_lds_kb=36
run --tuning roi_align_bwd_lds_kb=140
run --tuning roi_align_bwd_threads=256
run --tuning roi_align_bwd_threads=1024
