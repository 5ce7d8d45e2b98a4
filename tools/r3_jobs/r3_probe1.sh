#!/bin/bash
# round-3 data gathering: small-batch per-layer times, weight-gradient epilogue ablation, one- vs two-stream backward, batch-2 serialized kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p1
mkdir -p $OUT
cd $ROOT
python3 tools/bench_conv.py --set s3fd --batch 2 --which fwd,dgrad,wgrad > $OUT/conv_b2.txt 2>&1
python3 tools/bench_conv.py --set tail --batch 2 --which fwd,dgrad,wgrad >> $OUT/conv_b2.txt 2>&1
python3 tools/bench_conv.py --set s3fd --which wgrad > $OUT/wgrad_b16.txt 2>&1
DANHIP_WGRAD_ABLATE=1 python3 tools/bench_conv.py --set s3fd --which wgrad > $OUT/wgrad_b16_noepi.txt 2>&1
DANHIP_WGRAD_ABLATE=1 python3 tools/bench_conv.py --set s3fd --batch 2 --which wgrad > $OUT/wgrad_b2_noepi.txt 2>&1
for i in 1 2; do
  DANHIP_WGRAD_STREAM=1 python3 bench.py --steps 20 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | cut -c1-400 >> $OUT/stream_ab.txt
  DANHIP_WGRAD_STREAM=0 python3 bench.py --steps 20 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | cut -c1-400 >> $OUT/stream_ab.txt
done
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for b in 2 16; do
DANHIP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $OUT/serial_b$b -o s -- python3 bench.py --batch-per-gpu $b --steps 4 --warmup 2 --no-cpu-baseline --no-serialized-roofline --no-eval > $OUT/serial_b$b.log 2>&1
python3 tools/prof_db.py $OUT/serial_b$b/s_results.db 6 70 > $OUT/sfd_b${b}_serialized_kernels.txt
rm -rf $OUT/serial_b$b
done
ls -la $OUT
