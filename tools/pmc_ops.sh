#!/bin/bash
# scratch: PMC passes (each under its own timeout, never with tracing flags) for bench_ops sections
#   usage: tools/pmc_ops.sh <tag> <sections,comma> "<pmc set 1>" ["<pmc set 2>" ...]
TAG=$1; SEC=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY="import sys; sys.path.insert(0, '$ROOT'); import torch, bench_ops; torch.cuda.set_device(0); bench_ops.run(cpu=False, only=set('$SEC'.split(',')))"
i=0
for pmc in "$@"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $pmc --output-format csv -d $OUT/p$i -o pmc -- python -c "$PY" > $OUT/p$i.log 2>&1
done
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "sd::" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
