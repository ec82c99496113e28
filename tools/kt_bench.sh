#!/bin/bash
# Kernel-only durations (rocprofv3 --kernel-trace --stats) of the bench step for several knob sets.
#   (profiling build by default; KT_PRODUCT=1 times the product library)
#   usage: tools/kt_bench.sh "<bench flags 1>" "<bench flags 2>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
[ -z "$KT_PRODUCT" ] && [ -f $ROOT/tools/libsimpledet_ops_hip_prof.so ] && export SIMPLEDET_AMD_LIB=$ROOT/tools/libsimpledet_ops_hip_prof.so
cd /tmp && export TMPDIR=/tmp
i=0
for flags in "$@"; do
  i=$((i+1))
  rm -rf /tmp/ktb$i
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktb$i -o kt -- \
    python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --no-extra $flags > /tmp/ktb$i.log 2>&1
  echo "== $flags"
  python - <<PY
import csv, glob
for f in glob.glob("/tmp/ktb$i/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sd::" in r["Name"]:
            print("   %-70s calls %5s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
