#!/bin/bash
# scratch: PMC passes over the fused DCN forward alone (tools/dcn_fused_once.py)
#   usage: tools/dcn_fused_pmc.sh <tag> <sigma> "<pmc set 1>" ["<pmc set 2>" ...]
TAG=$1; SIG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "$@"; do
  i=$((i+1))
  ( cd $ROOT && timeout 120 rocprofv3 --pmc $pmc --output-format csv -d $OUT/p$i -o pmc -- python tools/dcn_fused_once.py $SIG > $OUT/p$i.log 2>&1 )
done
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "dcn_fwd_fused" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
