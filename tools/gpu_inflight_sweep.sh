#!/bin/bash
# bench.py value by (PS_CHAIN_T, --inflight): do kernels built for one workgroup per CU pay off when rollouts are pipelined?
cd "$(dirname "$0")/.."
for t in ${TS:-0 4 84}; do for n in ${NS:-2 3 4}; do
  v=$(PS_CHAIN_T=$t python bench.py --no-cpu-baseline --inflight $n 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f M  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))")
  echo "PS_CHAIN_T=$t inflight=$n: $v"
done; done
