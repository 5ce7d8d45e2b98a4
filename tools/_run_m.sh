timeout 600 python -m pytest tests/test_deform_form_gpu.py tests/test_deform_gpu.py tests/test_deform_variants_gpu.py -x -q -m gpu 2>&1 | tail -8
for sc in 0.3 0.6 1.0; do
  OFF_SCALE=$sc python tools/bench_deform_bwd.py 2>&1 | tail -1
done
