#!/bin/bash
# Dev (GPU box): where does the all-taps 3x3 weight gradient (k_conv_wgrad3) lose its matrix-pipe time? Times the layer with parts of the
# K loop compiled out (make -C openset-imagenet_amd/csrc ablate -> tools/probes/bin/libosi_hip_abl<bits>.so, -DOSI_ABLATE=<bits>: 1 no global loads in
# the loop, 2 no LDS stores, 4 no barriers, 8 no tap mask, 16 no LDS operand reads). Wrong results, right timing.
cd "$(dirname "$0")/../.."
echo "TFLOP/s, B = 128, wgrad + fused activation:   3x3 64@56  128@28  256@14  512@7  s2 128@56  s2 256@28"
for round in 1 2; do
for v in 0 1 2 4 7 8 15 31; do
  OSI_HIP_LIB=$PWD/tools/probes/bin/libosi_hip_abl$v.so python tools/time_wgrad.py || exit 1
done
done
