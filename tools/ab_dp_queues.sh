#!/bin/bash
# Same box, interleaved (round 4): does the data-parallel hand-off's cost come from hardware-queue aliasing? handoff = staged backward +
# osi_resnet50_grads_ready, RCCL's all_reduce patched out; dp = with it. "hi" / "normal" = priority of the communication stream
# (OSI_DP_COMM_PRIO), "q8" = GPU_MAX_HW_QUEUES=8. Result: profiles/r04_ab_dp_queues.txt
for r in 1 2; do
for cfg in plain "handoff hi" "handoff normal" "handoff hi q8" "dp hi" "dp normal" "dp hi q8"; do
  set -- $cfg
  e="X=1"; a="--force-dp"
  case $1 in plain) a="";; handoff) e="OSI_BENCH_SKIP_COLLECTIVE=1";; esac
  [ "$2" = normal ] && e="$e OSI_DP_COMM_PRIO=0"
  [ "$3" = q8 ] && e="$e GPU_MAX_HW_QUEUES=8"
  env $e python bench.py --no-cpu-baseline --no-profile $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', 'ms/step', d['ms_per_step'], d['windows_ms_per_step'])"
done; done
