#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p17
mkdir -p $OUT
cd $ROOT
for ab in 0 16; do
echo "== ablate $ab"
DANHIP_HALO2_ABLATE=$ab timeout 60 tools/halo2_trace fwd | grep -v "^ *[0-9]* tap"
DANHIP_HALO2=1 DANHIP_HALO2_ABLATE=$ab timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2 --check 2>&1 | grep -v amdgpu | cut -c1-110
done
DANHIP_HALO2=1 DANHIP_HALO2_ABLATE=16 timeout 600 python3 -m pytest tests/test_conv_gpu.py -q -m gpu -x -k "halo2" 2>&1 | tail -3
