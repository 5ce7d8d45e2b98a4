mkdir -p gpurun_out/j11
python -m pytest tests/test_conv_gpu.py -q -k "pool_only" 2>&1 | tail -n 3
for i in 1 2; do for v in 1 0; do DANHIP_POOL_ONLY=$v python tools/bench_eval_step.py 30 2>/dev/null | tail -n 1 | sed "s/^/pool_only=$v /"; done; done
