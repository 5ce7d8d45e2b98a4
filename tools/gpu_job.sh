mkdir -p gpurun_out/j13
python -m pytest tests/test_ops_gpu.py tests/test_conv_gpu.py tests/test_sfd_gpu.py -q -x 2>&1 | tail -n 6
python -m pytest tests/test_grad_parity_gpu.py tests/test_models_gpu.py tests/test_train_models_gpu.py tests/test_full_size_gpu.py -q -x -k "not 1024" 2>&1 | tail -n 4
for i in 1 2; do for v in 1 0; do DANHIP_POOL_ARG=$v python bench.py --steps 20 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pool_arg=$v', d['value'], d['ms_per_step'])"; done; done
