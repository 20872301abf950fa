#!/bin/bash
# Same-box, interleaved A/B of a committed tree against the working tree. Prepare with
#   rm -rf .ab/prev && mkdir -p .ab/prev && git archive <commit> bench.py openset-imagenet_amd oracle include config profiles/r03_hbm_traffic_per_step.json | tar -x -C .ab/prev && make -C .ab/prev/openset-imagenet_amd/csrc -j8
# then on the GPU box: tools/ab_tree.sh [rounds]   (each tree runs its OWN bench.py and library)
rounds=${1:-3}
for r in $(seq 1 $rounds); do
  for t in prev head; do
    if [ $t = head ]; then d=.; else d=.ab/$t; fi
    (cd $d && python bench.py --no-cpu-baseline 2>/dev/null) | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']
print('$t', 'ms/step', d['ms_per_step'], d.get('windows_ms_per_step', ''), 'serialized: fwd', pc['conv_fwd']['ms_per_step'], 'dgrad', pc['conv_dgrad']['ms_per_step'], 'wgrad', pc['conv_wgrad']['ms_per_step'], 'bn_fwd', pc['bn_fwd']['ms_per_step'], 'bn_bwd', pc['bn_bwd']['ms_per_step'], 'total', d['roofline']['serialized_ms_per_step'])"
  done
done
