#!/bin/bash
# A/B of kernel variants on one GPU box (scratch helper): each argument is a bench.py flag string
# ablation knobs live in the profiling build only (make -C simpledet_amd/csrc prof)
[ -f tools/libsimpledet_ops_hip_prof.so ] && export SIMPLEDET_AMD_LIB=$PWD/tools/libsimpledet_ops_hip_prof.so
run() { echo "== $*"; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops $* 2>/tmp/ab_err.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  step %.3f ms  fwd %.3f ms (%.1f%%)  bwd %.3f ms (%.1f%%)' % (d['ms_per_step'], r['avg_launch_ms'], 100*r['frac'], r['backward']['avg_ms'], 100*r['backward']['frac']))
"; }
for v in "$@"; do run $v; done
