#!/bin/bash
# rocprofv3 kernel-trace stats of selected bench_ops sections: tools/kt_ops.sh <tag> <sections,comma> [k=v ...]
TAG=$1; SEC=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import torch, bench_ops
from simpledet_amd._lib import lib
for kv in '$*'.split():
    k, v = kv.split('='); lib().set_tuning(k, int(v))
torch.cuda.set_device(0)
bench_ops.run(cpu=False, only=set('$SEC'.split(',')))" > $OUT/kt.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sd::" in r["Name"]:
            print("%-90s calls %5s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
