#!/bin/bash
# end-of-round evidence job (one gpurun call): profiles/r3 inputs
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/collect_round_profiles.sh r3c > gpurun_out/r3c_collect.log 2>&1
OUT=$ROOT/gpurun_out/r3c
python3 tools/dp_bucket_drift.py --model pb --steps 200 > $OUT/pb_bucket_drift.json 2> $OUT/pb_bucket_drift.err
for w in fwd dgrad_bits wgrad; do
  bash tools/pmc_conv.sh conv3_2 $w > $OUT/pmc_conv3_2_$w.txt 2>&1
done
ls -la $OUT
tail -3 $OUT/pb_bucket_drift.json | cut -c1-600
