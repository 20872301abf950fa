#!/bin/bash
# Same box, interleaved: the plain single-GPU step against the data-parallel code path at world size 1 (--force-dp: staged backward,
# hand-off to the communication stream, one RCCL all_reduce per stage on a world-1 communicator).  tools/ab_force_dp.sh [rounds]
rounds=${1:-3}
for r in $(seq 1 $rounds); do
  for v in plain force-dp; do
    if [ $v = plain ]; then a=""; else a="--force-dp"; fi
    python bench.py --no-cpu-baseline --no-profile $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d.get('rccl') or {}
print('$v', 'ms/step', d['ms_per_step'], d['windows_ms_per_step'], 'per_bucket_comm_ms', r.get('per_bucket_comm_ms'), 'exposed', r.get('exposed_comm_ms'), 'channels', (r.get('channels') or {}).get('coll_channels'))"
  done
done
