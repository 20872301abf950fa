#!/bin/bash
# Same box, interleaved (round 4): the ceiling of folding the forward's block-output pass into the next block's conv1. plain step; conv1
# recomputing the block output in its loader with the materialising pass beside it (OSI_FWD_RECOMPUTE=1, round 2's experiment); the same
# WITHOUT the pass (dbg_skip bit 2: wrong results, right timing = two-stream conv1, no pass, no stores); no block-output passes at all
# (dbg_skip bit 1). Result: profiles/r04_upper_bound_block_output_in_conv1.txt
make -C openset-imagenet_amd/csrc diag >/dev/null || exit 1
for r in 1 2; do
for cfg in "OSI_DBG_SKIP=0" "OSI_FWD_RECOMPUTE=1" "OSI_FWD_RECOMPUTE=1 OSI_DBG_SKIP=4" "OSI_DBG_SKIP=2"; do
  env OSI_DEV=1 OSI_HIP_LIB=$PWD/openset-imagenet_amd/csrc/libosi_hip_diag.so $cfg python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']
print('$cfg', 'ms/step', d['ms_per_step'], d['windows_ms_per_step'], 'fwd', pc['conv_fwd']['ms_per_step'], 'dgrad', pc['conv_dgrad']['ms_per_step'], 'wgrad', pc['conv_wgrad']['ms_per_step'], 'bn', pc['bn_fwd']['ms_per_step'], pc['bn_bwd']['ms_per_step'])"
done; done
