#!/bin/bash
# scratch: a few safe PMC passes (each under its own timeout) for a bench.py variant
#   usage: tools/pmc_quick.sh <tag> "<bench args>" "<pmc set 1>" ["<pmc set 2>" ...]
TAG=$1; BARGS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ops $BARGS"
i=0
for pmc in "$@"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $pmc --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "sd::" in row["Kernel_Name"] and "stream_copy" not in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:44]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
