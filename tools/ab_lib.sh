#!/bin/bash
# A/B of two BUILDS of the library on one GPU box, alternating: tools/ab_lib.sh <other.so> [pairs] [bench flags...]
OTHER=$1; PAIRS=${2:-3}; shift 2
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --no-extra $* 2>/tmp/ab_err.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('  step %.4f ms  fwd %.4f ms  bwd %.4f ms' % (d['ms_per_step'], r['avg_launch_ms'], r['backward']['avg_ms']))
"; }
for i in $(seq $PAIRS); do
  echo "== shipped"; run $*
  echo "== $OTHER"; SIMPLEDET_AMD_LIB=$PWD/$OTHER run $*
done
