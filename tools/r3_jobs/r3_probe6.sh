#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p6
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests/test_grad_parity_gpu.py tests/test_conv_gpu.py -q -m gpu -k "imposed or slab or first_layer" > $OUT/tests.txt 2>&1
tail -6 $OUT/tests.txt
for i in 1 2; do
python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask --only conv2_2,conv3_1,conv3_2 --check >> $OUT/halo_base.txt 2>&1
DANHIP_HALO_TPS3=1 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask --only conv2_2,conv3_1,conv3_2 --check >> $OUT/halo_tps3.txt 2>&1
done
paste $OUT/halo_base.txt $OUT/halo_tps3.txt | cut -c1-60,130-190
python3 tools/bench_conv.py --set s3fd --which wgrad --batch 2 > $OUT/wgrad_b2.txt 2>&1; cat $OUT/wgrad_b2.txt
