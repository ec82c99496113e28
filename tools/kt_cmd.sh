#!/bin/bash
# kernel-only durations of an arbitrary command: tools/kt_cmd.sh <tag> <command...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG
( cd $ROOT && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o kt -- "$@" > /tmp/kt_$TAG.log 2>&1 )
tail -3 /tmp/kt_$TAG.log
python - <<PY
import csv, glob
for f in glob.glob("/tmp/kt_$TAG/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sd::" in r["Name"] or "sd_" in r["Name"]:
            print("   %-90s calls %5s avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
