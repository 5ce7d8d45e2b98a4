#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p4
mkdir -p $OUT
cd $ROOT
timeout 1200 python3 -m pytest tests/test_conv_gpu.py tests/test_deform_gpu.py tests/test_concat_gpu.py tests/test_grad_parity_gpu.py tests/test_ops_gpu.py tests/test_sfd_gpu.py tests/test_train_models_gpu.py -q -m gpu -x > $OUT/tests.txt 2>&1
tail -12 $OUT/tests.txt
python3 tools/bench_conv.py --set s3fd --which fwd,dgrad,wgrad --only conv1_1,conv1_2 > $OUT/conv_first.txt 2>&1
cat $OUT/conv_first.txt
for b in 2 16; do
  python3 bench.py --batch-per-gpu $b --steps 20 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | cut -c1-520 >> $OUT/bench_lines.txt
done
cut -c60-200 $OUT/bench_lines.txt
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for b in 2 16; do
DANHIP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $OUT/serial_b$b -o s -- python3 bench.py --eager --batch-per-gpu $b --steps 4 --warmup 2 --no-cpu-baseline --no-serialized-roofline --no-eval > $OUT/serial_b$b.log 2>&1
python3 tools/prof_db.py $OUT/serial_b$b/s_results.db 6 45 > $OUT/sfd_b${b}_serialized_kernels.txt
rm -rf $OUT/serial_b$b
done
