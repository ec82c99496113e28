#!/bin/bash
# Per-round profile of the bench command on the GPU box.  Everything lands in gpurun_out/<tag>/
# (gpurun merges that directory back); the summaries are then copied into profiles/ and committed.
#   usage: tools/profile_round.sh <tag> [bench args...]
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ops --calibrate $*"
# 1. kernel trace + stats (average duration per kernel)
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
# 2. PMC passes, one counter group per run (never combined with tracing)
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc$i -o pmc -- $CMD > $OUT/pmc$i.log 2>&1
done
cd $ROOT
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
