#!/bin/bash
# Same box, interleaved (round 6): what would removing the forward's 65 BatchNorm finalize launches buy at most?
# Diagnostic build, dbg_skip bit 3 (the launches are simply not issued: wrong results, right timing), at B = 64 and B = 128.
# Result: profiles/r06_upper_bound_finalize_launches.txt
# The diagnostic build is swapped in place of libosi_hip.so for the run (the op library links that name: ONE instance of the C ABI in the process).
make -C openset-imagenet_amd/csrc diag >/dev/null || exit 1
L=openset-imagenet_amd/csrc/libosi_hip.so
cp $L /tmp/libosi_hip_keep.so && cp openset-imagenet_amd/csrc/libosi_hip_diag.so $L || exit 1
trap 'cp /tmp/libosi_hip_keep.so '$L EXIT
for r in 1 2 3; do for B in 64 128; do for cfg in "OSI_DBG_SKIP=0" "OSI_DBG_SKIP=8"; do
  env OSI_DEV=1 OSI_HIP_LIB=$PWD/$L $cfg python bench.py --batch $B --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=$B $cfg', 'ms/step', d['ms_per_step'], d['windows_ms_per_step'], 'img/s', d['value'])"
done; done; done
