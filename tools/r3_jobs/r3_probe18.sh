#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p18
mkdir -p $OUT
cd $ROOT
for i in 1 2; do
for b in 1 0; do
echo "== halo_b2=$b"
DANHIP_HALO_B2=$b timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2,conv4_1,conv4_2 --check 2>&1 | grep -v amdgpu | cut -c1-64
DANHIP_HALO_B2=$b timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask --only conv2_1,conv5_1 2>&1 | grep -v amdgpu | cut -c1-64
done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
DANHIP_HALO_B2=0 timeout 900 python3 -m pytest tests/test_conv_gpu.py -q -m gpu -x 2>&1 | tail -3
