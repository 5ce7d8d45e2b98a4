#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p23
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests/test_conv_gpu.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
for lib in dan_amd/libdanhip_prev.so dan_amd/libdanhip.so; do
echo "== $lib"
DANHIP_LIB_PATH=$lib timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2,conv4_1,conv4_2 2>&1 | grep -v amdgpu | cut -c1-64
DANHIP_LIB_PATH=$lib timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask --only conv2_1 2>&1 | grep -v amdgpu | grep -v TOTAL| cut -c1-64
done
done > $OUT/ab.txt 2>&1
grep "==\|TOTAL\|conv2_1" $OUT/ab.txt
