#!/bin/bash
# scratch: PMC passes focused on the forward kernel's memory pipeline
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcf
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TD|TCP|SQ|TCC)_[A-Z0-9_]+" | sort -u > $OUT/counters.txt
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ops $BENCH_ARGS"
i=0
for pmc in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmcf"
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "roi_align_fwd" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            vg = row["VGPR_Count"], row["LDS_Block_Size"], row["SGPR_Count"]
    for k, cs in agg.items():
        print(k, vg, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
