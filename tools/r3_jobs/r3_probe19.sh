#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p19
mkdir -p $OUT
cd $ROOT
for i in 1 2; do
for b in 1 0; do
echo "== wgrad_b2=$b"
DANHIP_WGRAD_B2=$b timeout 300 python3 tools/bench_conv.py --set s3fd --which wgrad --check 2>&1 | grep -v amdgpu | cut -c1-100
done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
DANHIP_WGRAD_B2=0 DANHIP_HALO_B2=0 timeout 900 python3 -m pytest tests/test_conv_gpu.py -q -m gpu -x 2>&1 | tail -3
