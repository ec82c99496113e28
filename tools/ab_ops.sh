#!/bin/bash
# A/B of the secondary ops on one GPU box: tools/ab_ops.sh <sections,comma> ["k=v k=v" ...]
SEC=$1; shift
run() { echo "== tuning: $*"; python - "$SEC" $* <<'PY'
import sys, json
sys.path.insert(0, ".")
import torch, bench_ops
from simpledet_amd._lib import lib
for kv in sys.argv[2:]:
    k, v = kv.split("="); lib().set_tuning(k, int(v))
r = bench_ops.run(cpu=False, only=set(sys.argv[1].split(",")))
for k, v in r.items():
    print("  ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "config"})
PY
}
if [ $# -eq 0 ]; then run; else for v in "$@"; do run $v; done; fi
