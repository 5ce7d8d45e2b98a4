#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p25
mkdir -p $OUT
cd $ROOT
for i in 1 2; do
for h in 0 1; do
echo "== halo2=$h"
DANHIP_HALO2=$h timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2 2>&1 | grep -v amdgpu | cut -c1-64
done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
