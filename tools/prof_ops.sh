#!/bin/bash
# kernel-trace stats of the secondary ops (bench_ops.run without the CPU legs)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ops}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import torch, bench_ops
torch.cuda.set_device(0)
bench_ops.run(cpu=False)" > $OUT/kt.log 2>&1
python - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/kt/kt_kernel_stats.csv"))):
    if "sd::" in r["Name"]:
        print("%-100s calls %5s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
