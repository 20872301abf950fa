#!/bin/bash
# Dev (GPU box): the Winograd stream-K modes (wino_streamk 1 = both directions, 2 = forward only, 0 = never) at the reference's default batch 64, interleaved.
for r in 1 2 3; do for v in 1 2 0; do
OSI_WINO_STREAMK=$v python bench.py --batch 64 --no-cpu-baseline --sustained-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']
print('B=64 OSI_WINO_STREAMK=$v', 'ms/step', d['ms_per_step'], 'fwd', pc['conv_fwd']['ms_per_step'], 'dgrad', pc['conv_dgrad']['ms_per_step'], 'wgrad', pc['conv_wgrad']['ms_per_step'])"
done; done
