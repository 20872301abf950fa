#!/bin/bash
# Same box, interleaved: the step with the Winograd forms of the 3x3 stride-1 layers on / off (forward, input gradient).
# Result: profiles/r05_ab_wino.txt
for r in 1 2; do
for cfg in "OSI_FWD_WINO=0 OSI_DGRAD_WINO=0" "OSI_FWD_WINO=1 OSI_DGRAD_WINO=0" "OSI_FWD_WINO=0 OSI_DGRAD_WINO=1" "OSI_FWD_WINO=1 OSI_DGRAD_WINO=1"; do
  env $cfg python bench.py --no-cpu-baseline --sustained-steps 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']
print('$cfg', 'ms/step', d['ms_per_step'], d['windows_ms_per_step'], 'img/s', d['value'], 'fwd', pc['conv_fwd']['ms_per_step'], 'dgrad', pc['conv_dgrad']['ms_per_step'], 'wgrad', pc['conv_wgrad']['ms_per_step'], 'bn', pc['bn_fwd']['ms_per_step'], pc['bn_bwd']['ms_per_step'], 'conv frac', d['roofline']['frac'], 'loss', d['final_loss'])"
done; done
