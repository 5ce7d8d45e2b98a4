#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p15
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests/test_conv_gpu.py tests/test_ops_gpu.py -q -m gpu -x > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
for i in 1 2; do
DANHIP_HALO2=0 timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask 2>&1 | grep -v amdgpu >> $OUT/halo1.txt
DANHIP_HALO2=0 timeout 300 python3 tools/bench_conv.py --set s3fd --which dgrad_bits --only conv2_2,conv3_1,conv3_2,conv4_1,conv4_2,conv5_1 2>&1 | grep -v amdgpu >> $OUT/halo1.txt
done
cut -c1-100 $OUT/halo1.txt
DANHIP_HALO2=1 timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2 2>&1 | grep -v amdgpu | cut -c1-100 > $OUT/halo2.txt
cat $OUT/halo2.txt
