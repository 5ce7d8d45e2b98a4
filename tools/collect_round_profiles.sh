#!/bin/bash
# Collects the per-round evidence under gpurun_out/<tag>/ (copy what is kept into profiles/<round>/):
#   bench line, rocprofv3 kernel stats of the same command, PMC traffic, the other graphs' lines, the small-batch (strong-scaling shard) lines.
# usage (on the GPU box): bash tools/collect_round_profiles.sh c
set -u
TAG=${1:-c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
# PMC traffic FIRST: bench.py reports roofline.traffic only from a file stamped with the hash of the sources it runs on
bash tools/pmc_bench.sh r6 > $OUT/pmc.log 2>&1; mkdir -p profiles/r6; cp gpurun_out/pmc_bench_traffic.json profiles/r6/pmc_bench_traffic.json
cd $ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/s3fd_b16_bench_line.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o s3fd -- python3 bench.py --steps 5 --warmup 2 --repeats 1 --strong-global-batch 0 --strong16-global-batch 0 --no-cpu-baseline --no-serialized-roofline --no-eval > $OUT/s3fd_b16_bench_line_under_rocprof.json 2> $OUT/prof.err
cp $OUT/prof/*kernel_stats.csv $OUT/s3fd_b16_kernel_stats.csv 2>/dev/null || find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/s3fd_b16_kernel_stats.csv \;
python3 tools/timeline.py $(find $OUT/prof -name "*kernel_trace.csv" | tail -1) 150 > $OUT/s3fd_b16_step_timeline.txt 2>/dev/null
# SQ_VALU_MFMA_BUSY_CYCLES (+ LDS / wait counters, traffic) of the three 3x3 kernel families on conv3_2 and of the 16 x 16-tile forward (missing 5)
for job in "conv3_2 fwd" "conv3_2 dgrad_bits" "conv3_2 wgrad" "conv4_2 fwd" "conv4_2 dgrad_bits"; do
  set -- $job
  bash tools/pmc_conv.sh $1 $2 > $OUT/pmc_$1_$2.txt 2>&1
done
cp gpurun_out/pmc_bench_traffic.json $OUT/ 2>/dev/null
: > $OUT/models_bench_lines.jsonl
for m in pb dan dan_deform; do
  python3 bench.py --model $m --steps 30 --repeats 1 --strong-global-batch 0 --strong16-global-batch 0 --no-cpu-baseline --no-serialized-roofline 2>/dev/null | tail -1 >> $OUT/models_bench_lines.jsonl
  python3 bench.py --model $m --steps 30 --repeats 1 --graph --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 >> $OUT/models_bench_lines.jsonl
done
python3 bench.py --steps 30 --repeats 1 --graph --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 >> $OUT/models_bench_lines.jsonl
: > $OUT/s3fd_small_batch_lines.jsonl
for b in 2 4 8; do
  python3 bench.py --batch-per-gpu $b --steps 40 --repeats 1 --strong-global-batch 0 --strong16-global-batch 0 --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 >> $OUT/s3fd_small_batch_lines.jsonl
done
# BASELINE.json configs[3] / [4] at their input size: per-GPU shards (batch 8 at 1024 x 1024) of DAN (bf16) and DAN-Deform (fp16 build)
: > $OUT/size1024_lines.jsonl
for mode in --eager --graph; do
  python3 bench.py --model dan --size 1024 --batch-per-gpu 8 --steps 5 --repeats 1 --strong-global-batch 0 --strong16-global-batch 0 $mode --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 >> $OUT/size1024_lines.jsonl
  DANHIP_DTYPE=fp16 python3 bench.py --model dan_deform --size 1024 --batch-per-gpu 8 --steps 5 --repeats 1 --strong-global-batch 0 --strong16-global-batch 0 $mode --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 >> $OUT/size1024_lines.jsonl
done
DANHIP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $OUT/serial_dan1024 -o s -- python3 bench.py --eager --model dan --size 1024 --batch-per-gpu 8 --steps 3 --warmup 1 --repeats 1 --strong-global-batch 0 --strong16-global-batch 0 --no-cpu-baseline --no-serialized-roofline --no-eval > $OUT/serial_dan1024.log 2>&1
python3 tools/prof_db.py $OUT/serial_dan1024/s_results.db 5 40 > $OUT/dan_1024_b8_serialized_kernels.txt
rm -rf $OUT/serial_dan1024
DANHIP_DTYPE=fp16 DANHIP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $OUT/serial_dd1024 -o s -- python3 bench.py --eager --model dan_deform --size 1024 --batch-per-gpu 8 --steps 3 --warmup 1 --repeats 1 --strong-global-batch 0 --strong16-global-batch 0 --no-cpu-baseline --no-serialized-roofline --no-eval > $OUT/serial_dd1024.log 2>&1
python3 tools/prof_db.py $OUT/serial_dd1024/s_results.db 5 40 > $OUT/dan_deform_fp16_1024_b8_serialized_kernels.txt
rm -rf $OUT/serial_dd1024
# the per-rank shape of an 8-GPU strong-scaling run (2 images per GPU): serialized kernel list
DANHIP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $OUT/serial_b2 -o s -- python3 bench.py --eager --batch-per-gpu 2 --steps 4 --warmup 2 --repeats 1 --strong-global-batch 0 --strong16-global-batch 0 --no-cpu-baseline --no-serialized-roofline --no-eval > $OUT/serial_b2.log 2>&1
python3 tools/prof_db.py $OUT/serial_b2/s_results.db 7 50 > $OUT/sfd_b2_serialized_kernels.txt
rm -rf $OUT/serial_b2
for m in sfd dan dan_deform pb; do
  DANHIP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $OUT/serial_$m -o s -- python3 bench.py --eager --model $m --steps 4 --warmup 2 --repeats 1 --strong-global-batch 0 --strong16-global-batch 0 --no-cpu-baseline --no-serialized-roofline --no-eval > $OUT/serial_$m.log 2>&1
  python3 tools/prof_db.py $OUT/serial_$m/s_results.db 7 60 > $OUT/${m}_b16_serialized_kernels.txt
  rm -rf $OUT/serial_$m
done
# event-bracketed per-layer convolution times of one eager DAN step (one stream) and the torch-native launches left on it
DANHIP_WGRAD_STREAM=0 python3 tools/host_time.py dan layers 2>/dev/null | grep -v "^dan:" > $OUT/dan_b16_layer_times.txt
python3 tools/host_time.py dan aten 2>/dev/null | grep " x aten\|per step\| x danhip" > $OUT/dan_b16_torch_native_launches.txt
# the test-time pipeline at 1024 x 768 (eval scripts without a trainer: fused blocks cached per weight version)
timeout 900 python3 tools/bench_eval.py 2>/dev/null > $OUT/eval_pipeline_1024x768.txt
rm -rf $OUT/prof
ls -la $OUT
