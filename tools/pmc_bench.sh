#!/bin/bash
# scratch: PMC passes over the bench command (no tracing flags): tools/pmc_bench.sh <tag> "<pmc set>" ...
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "$@"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $pmc --output-format csv -d $OUT/p$i -o pmc -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops --no-extra > $OUT/p$i.log 2>&1
done
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "sd::" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
