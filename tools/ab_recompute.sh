for r in 1 2; do
for cfg in "OSI_DBG_SKIP=0" "OSI_FWD_RECOMPUTE=1" "OSI_FWD_RECOMPUTE=1 OSI_DBG_SKIP=4" "OSI_DBG_SKIP=2"; do
  env $cfg python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']
print('$cfg', 'ms/step', d['ms_per_step'], d['windows_ms_per_step'], 'fwd', pc['conv_fwd']['ms_per_step'], 'dgrad', pc['conv_dgrad']['ms_per_step'], 'wgrad', pc['conv_wgrad']['ms_per_step'], 'bn', pc['bn_fwd']['ms_per_step'], pc['bn_bwd']['ms_per_step'])"
done; done
