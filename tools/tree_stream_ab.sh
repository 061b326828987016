#!/bin/bash
# On the GPU box: the default bench (short form, with the bounded Go samples) three times in one call - tree / scan / gather launches on
# the engine's own stream (product), on a second stream of the engine (SPRL_TREE_STREAM=1), on a second stream at the device's highest
# priority (=2).  usage: tools/tree_stream_ab.sh <prefix>
set -o pipefail
PREFIX=$1
cd "$(dirname "$0")/.."
for v in 0 2 1 0; do
    if [ $v = 0 ]; then unset SPRL_TREE_STREAM; else export SPRL_TREE_STREAM=$v; fi
    timeout -k 10 400 python3 bench.py --allow-lab --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/${PREFIX}_bench_tree_stream_$v.json 2> gpurun_out/${PREFIX}_bench_tree_stream_$v.err || { tail -5 gpurun_out/${PREFIX}_bench_tree_stream_$v.err; exit 1; }
    python3 -c "
import json
d=json.loads(open('gpurun_out/${PREFIX}_bench_tree_stream_$v.json').read().strip().splitlines()[-1])
s=d.get('secondary',{})
print('SPRL_TREE_STREAM=$v', 'othello', round(d['value'],1), 'games/s;', 'go9', round(s['go9']['nn_evals_per_sec']/1e6,3), 'M evals/s; go19', round(s['go19']['nn_evals_per_sec']/1e6,3), 'M evals/s;', d['config']['evaluator'][-60:])"
done
