#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) for a list of tuning settings
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for t in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/abk$i -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline $t > /dev/null 2>&1
  echo "== $t"; python - <<PY
import csv
rows=list(csv.reader(open("$OUT/abk$i/kt_kernel_stats.csv")))
for r in rows[1:5]:
    if 'roi_align' in r[0]: print("   %-62s calls %s avg %.1f us min %.1f max %.1f" % (r[0][:62], r[1], float(r[3])/1e3, float(r[5])/1e3, float(r[6])/1e3))
PY
done
