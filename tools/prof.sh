#!/bin/bash
# scratch profiling helper: kernel trace + a few PMC passes; everything lands in gpurun_out/
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > /dev/null 2>&1
i=0
for pmc in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc$i -o pmc -- $CMD > $OUT/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    print("==", f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 12: print("  ", [c[:60] for c in row[:8]])
for d in sorted(glob.glob(out + "/pmc*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", f)
        for k, cs in agg.items():
            if "roi_align" not in k: continue
            print("  ", k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
