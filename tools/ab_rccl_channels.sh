#!/bin/bash
# Same box, interleaved: what RCCL's resident channel workgroups cost the step they run beside. World-1 communicator (--force-dp), the
# all_reduce kernels really run (61 / 28 / 5 / 1 MB in place). tools/ab_rccl_channels.sh "default 2 4 8 16" [rounds]
vals=${1:-"default 2 8"}; rounds=${2:-2}
python bench.py --no-cpu-baseline --no-profile --steps 3 --warmup 1 --windows 1 --force-dp > /dev/null 2> /tmp/osi_rccl_dbg.err; grep -h -i "channel" /tmp/osi_rccl_init_*.log 2>/dev/null | head -5
for r in $(seq 1 $rounds); do
  for v in plain $vals; do
    if [ $v = plain ]; then e=""; a=""; elif [ $v = default ]; then e=""; a="--force-dp"; else e="NCCL_MIN_NCHANNELS=$v NCCL_MAX_NCHANNELS=$v"; a="--force-dp"; fi
    env $e python bench.py --no-cpu-baseline --no-profile $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d.get('rccl') or {}
print('$v', 'ms/step', d['ms_per_step'], d['windows_ms_per_step'], 'per_bucket_comm_ms', r.get('per_bucket_comm_ms'), 'exposed', r.get('exposed_comm_ms'), 'channels', r.get('channels'))"
  done
done
