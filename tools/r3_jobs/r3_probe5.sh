#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p5
mkdir -p $OUT
cd $ROOT
timeout 1500 python3 -m pytest tests/test_conv_gpu.py tests/test_grad_parity_gpu.py tests/test_anchors_gpu.py tests/test_models_gpu.py tests/test_routing_gpu.py tests/test_eval_f32_gpu.py tests/test_evalpipe_gpu.py tests/test_ops_gpu.py tests/test_train_models_gpu.py -q -m gpu > $OUT/tests.txt 2>&1
tail -8 $OUT/tests.txt
for m in pb dan dan_deform; do
  for st in 1 0; do
    DANHIP_WGRAD_STREAM=$st python3 bench.py --model $m --steps 10 --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 | cut -c60-140 | sed "s/^/$m stream=$st eager /" >> $OUT/models.txt
    DANHIP_WGRAD_STREAM=$st python3 bench.py --model $m --steps 10 --graph --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 | cut -c60-140 | sed "s/^/$m stream=$st graph /" >> $OUT/models.txt
  done
done
cat $OUT/models.txt
for b in 2 4 16; do
  python3 bench.py --batch-per-gpu $b --steps 20 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | cut -c60-200 >> $OUT/bench_lines.txt
done
DANHIP_WGRAD_SLAB=0 python3 bench.py --batch-per-gpu 16 --steps 20 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | cut -c60-200 >> $OUT/bench_lines.txt
DANHIP_SPLITK=0 python3 bench.py --batch-per-gpu 16 --steps 20 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | cut -c60-200 >> $OUT/bench_lines.txt
cat $OUT/bench_lines.txt
