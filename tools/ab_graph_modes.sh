#!/bin/bash
# Same-box comparison of the step's launch modes: eager (two backward streams) against hipGraph replay under the HIP runtime's graph knobs
# (DEBUG_CLR_GRAPH_PACKET_CAPTURE = 0: branches of a captured graph run on several streams instead of one AQL packet batch;
# DEBUG_HIP_FORCE_GRAPH_QUEUES = n: number of those streams).  usage: bash tools/ab_graph_modes.sh [rounds] [extra bench args]
R=${1:-2}; shift
COMMON="--steps 40 --no-cpu-baseline --no-eval --strong-global-batch 0 --strong16-global-batch 0 --no-serialized-roofline --events-steps 1 --repeats 2 $@"
run() { name=$1; shift; out=$(env "$@" python3 bench.py $COMMON $EXTRA 2>/dev/null | tail -1); python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('%-28s %s  %s' % (sys.argv[2], d['repeats']['ms_per_step'], d['config']['step_launch']))" "$out" "$name"; }
for r in $(seq 1 $R); do
  EXTRA="--eager" run eager X=1
  EXTRA="--graph" run graph X=1
  EXTRA="--graph" run graph_packet0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  EXTRA="--graph" run graph_queues2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
  EXTRA="--graph" run graph_packet0_queues2 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
  EXTRA="--graph" run graph_packet0_queues4 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
done
