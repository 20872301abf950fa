#!/bin/bash
# Same-box, interleaved A/B of the committed end states of round 1 and round 2 against the working tree: prices on ONE box what each
# round's restructuring bought (the fused in-block activation of round 2: conv_fwd + / bn_fwd -). The old trees are exported with
# `git archive <round commit> bench.py openset-imagenet_amd oracle include config` into .ab/r1, .ab/r2 and built there (see
# profiles/NOTES_r03.md); each runs its OWN bench.py.   tools/ab_rounds.sh [rounds]
rounds=${1:-3}
for r in $(seq 1 $rounds); do
  for t in r1 r2 head; do
    if [ $t = head ]; then d=.; else d=.ab/$t; fi
    (cd $d && python bench.py --no-cpu-baseline 2>/dev/null) | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']
print('$t', 'ms/step', d['ms_per_step'], d.get('windows_ms_per_step', ''), 'serialized: fwd', pc['conv_fwd']['ms_per_step'], 'dgrad', pc['conv_dgrad']['ms_per_step'], 'wgrad', pc['conv_wgrad']['ms_per_step'], 'bn_fwd', pc['bn_fwd']['ms_per_step'], 'bn_bwd', pc['bn_bwd']['ms_per_step'], 'total', d['roofline']['serialized_ms_per_step'])"
  done
done
