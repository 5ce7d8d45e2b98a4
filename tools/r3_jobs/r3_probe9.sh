#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p9
mkdir -p $OUT
cd $ROOT
for ab in 0 1 2 4 8 3 15; do
  echo "== halo2 ablate=$ab" >> $OUT/ablate.txt
  DANHIP_HALO2=1 DANHIP_HALO2_ABLATE=$ab timeout 200 python3 tools/bench_conv.py --set s3fd --which fwd --only conv2_2,conv3_2 2>&1 | grep "fwd " >> $OUT/ablate.txt
done
echo "== halo1" >> $OUT/ablate.txt
DANHIP_HALO2=0 timeout 200 python3 tools/bench_conv.py --set s3fd --which fwd --only conv2_2,conv3_2 2>&1 | grep "fwd " >> $OUT/ablate.txt
cut -c1-70 $OUT/ablate.txt
