#!/bin/bash
# Batch generality: img/s, ms/step and the conv-stack fraction of `bench.py --batch B` for the reference's own default
# (config/train.yaml:18: 64), the benchmarked 128, Protocol 3's 256, and a ragged last batch (37: no drop_last, train.py:299-304).
# Result: profiles/r05_batch_sweep.txt
echo "B  img/s  ms/step  conv_frac  fwd_TF dgrad_TF wgrad_TF  bn_fwd_ms bn_bwd_ms"
for B in 32 64 96 128 192 256 37; do
  python bench.py --batch $B --no-cpu-baseline --sustained-steps 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; pc=r['per_class']
print($B, d['value'], d['ms_per_step'], r['frac'], pc['conv_fwd']['tflops'], pc['conv_dgrad']['tflops'], pc['conv_wgrad']['tflops'], pc['bn_fwd']['ms_per_step'], pc['bn_bwd']['ms_per_step'])"
done
