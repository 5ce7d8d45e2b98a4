#!/bin/bash
# alternating: round-4 tree vs current tree, models dan and sfd (+ opt overlap off)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
for r in 1 2 3; do
  for m in sfd dan; do
    (cd _r4tree && python bench.py --model $m --steps 30 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r4 ', '$m', d['value'], d['ms_per_step'])")
    python bench.py --model $m --steps 30 --repeats 1 --strong-global-batch 0 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r5 ', '$m', d['value'], d['ms_per_step'])"
    DANHIP_OPT_OVERLAP=0 python bench.py --model $m --steps 30 --repeats 1 --strong-global-batch 0 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r5-noovl', '$m', d['value'], d['ms_per_step'])"
  done
done
