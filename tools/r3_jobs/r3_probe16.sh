#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p16
mkdir -p $OUT
cd $ROOT
for i in 1 2 3; do
DANHIP_HALO2=0 timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2,conv4_1,conv4_2 2>&1 | grep -v amdgpu | cut -c1-60 >> $OUT/halo1.txt
done
cat $OUT/halo1.txt
