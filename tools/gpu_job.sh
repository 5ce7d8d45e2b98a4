mkdir -p gpurun_out/j16
for i in 1 2; do for v in 1 0; do DANHIP_FUSED_CONTEXT=$v python bench.py --model dan --steps 10 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dan eager fused=$v', d['value'], d['ms_per_step'])"; done; done
python __graft_entry__.py smoke 2>&1 | tail -n 2
python bench.py > gpurun_out/j16/bench_sfd.json 2>gpurun_out/j16/bench_sfd.err; cut -c1-200 gpurun_out/j16/bench_sfd.json
