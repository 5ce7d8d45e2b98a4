#!/bin/bash
# Same-box regression A/B against the previous round's tree: alternating processes, S3FD and DAN, plus this round's tree with the bucket-wise
# optimizer switched on.  Prepare on the build host (the tree travels with the snapshot; remove it afterwards):
#   git worktree add -f _r4tree <round-4 commit> && (cd _r4tree && python -m dan_amd.build)
#   gpurun -- 'bash tools/ab_vs_r4.sh'          ->  profiles/r5/ab_vs_round4_same_box.txt
#   git worktree remove --force _r4tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
for r in 1 2 3; do
  for m in sfd dan; do
    (cd _r4tree && python bench.py --model $m --steps 30 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r4 ', '$m', d['value'], d['ms_per_step'])")
    python bench.py --model $m --steps 30 --repeats 1 --strong-global-batch 0 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r5 ', '$m', d['value'], d['ms_per_step'])"
    DANHIP_OPT_OVERLAP=1 python bench.py --model $m --steps 30 --repeats 1 --strong-global-batch 0 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r5+ovl', '$m', d['value'], d['ms_per_step'])"
  done
done
