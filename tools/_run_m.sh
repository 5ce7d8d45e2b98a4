timeout 300 python -m pytest tests/test_deform_gpu.py tests/test_deform_variants_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/bench_deform_fwd.py 160 0.5 2>&1 | tail -5
python tools/bench_deform_fwd.py 160 0.0 2>&1 | tail -3
