#!/bin/bash
# Dev (GPU box): what reserving CUs for RCCL's resident channel workgroups costs the step at world size 1 (--force-dp): the direct
# kernels' tail plans and the Winograd kernels' persistent grids shrink by the reserved count.   tools/ab_dp_reserved.sh [rounds]
R=${1:-2}
for r in $(seq $R); do for v in 0 4 16 32; do
OSI_DP_RESERVED_CUS=$v python bench.py --force-dp --no-cpu-baseline --sustained-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pc=d['roofline']['per_class']
print('OSI_DP_RESERVED_CUS=$v', 'ms/step', d['ms_per_step'], 'fwd', pc['conv_fwd']['ms_per_step'], 'dgrad', pc['conv_dgrad']['ms_per_step'], 'wgrad', pc['conv_wgrad']['ms_per_step'], 'plan', d['rccl']['launch_plan']['tail_plan_cus_in_effect'])"
done; done
