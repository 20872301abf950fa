#!/bin/bash
# Same box, interleaved: the parts of the data-parallel step's overhead at world size 1. plain = one backward call; staged = four stage
# calls (stage_join 0), nothing else; handoff = + osi_resnet50_grads_ready per stage and the final wait for the communication stream;
# dp = + RCCL's all_reduce kernels (world-1 communicator).   tools/ab_dp_parts.sh [rounds]
for r in $(seq 1 ${1:-2}); do
for cfg in plain staged handoff dp; do
  case $cfg in plain) e="X=1"; a="";; staged) e="OSI_BENCH_SKIP_COLLECTIVE=2"; a="--force-dp";; handoff) e="OSI_BENCH_SKIP_COLLECTIVE=1"; a="--force-dp";; dp) e="X=1"; a="--force-dp";; esac
  env $e python bench.py --no-cpu-baseline --no-profile $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', 'ms/step', d['ms_per_step'], d['windows_ms_per_step'])"
done; done
