#!/bin/bash
# Profile of the secondary ops (bench_ops.run without the CPU legs): kernel-trace stats + PMC passes
# (FETCH_SIZE, WRITE_SIZE, L2 hit/miss), one counter group per run, never combined with tracing.
#   usage: tools/prof_ops.sh <tag>      -> gpurun_out/<tag>/{kernel_stats.csv,pmc_summary.json,summary.txt}
TAG=${1:-ops}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY="import sys; sys.path.insert(0, '$ROOT'); import torch, bench_ops; torch.cuda.set_device(0); bench_ops.run(cpu=False)"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python -c "$PY" > $OUT/kt.log 2>&1
i=0
# (round 5: the SQ passes -- VALU / LDS instruction and busy counters, LDS conflicts, wave cycles -- say what a
# kernel that is not HBM bound is bound by: DESIGN 4.3 quotes them for the fp16 14x14 forward and for soft-NMS)
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc$i -o pmc -- python -c "$PY" > $OUT/pmc$i.log 2>&1
done
cd $ROOT
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
