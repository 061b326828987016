#!/bin/bash
# On the GPU box: tools/bench_variants.sh <prefix> <variant> [<variant> ...] [-- bench args]: bench.py once per plugin variant built by
# tools/build_plugin_variant.sh (the variant is copied over sprl_amd/libsprl_amd_torch.so of the scratch snapshot), one JSON line each
# in gpurun_out/<prefix>_bench_<variant>.json.  Stops at the first failure.
set -o pipefail
PREFIX=$1; shift
VARS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARS+=("$1"); shift; done
[ "$1" == "--" ] && shift
ARGS=${@:---no-cpu-baseline --no-secondary --steps 10 --warmup 3}
cd "$(dirname "$0")/.."
for v in "${VARS[@]}"; do
    cp tools/variants/$v/libsprl_amd_torch.so sprl_amd/libsprl_amd_torch.so || exit 1
    timeout -k 10 500 python3 bench.py $ARGS > gpurun_out/${PREFIX}_bench_$v.json 2> gpurun_out/${PREFIX}_bench_$v.err || { tail -5 gpurun_out/${PREFIX}_bench_$v.err; exit 1; }
    python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${PREFIX}_bench_$v.json').read().strip().splitlines()[-1])
print('$v', round(d['value'],1), d['unit'], 'conv frac', round(d['roofline']['frac'],3), 'ms/step', round(d['ms_per_step'],1))"
done
