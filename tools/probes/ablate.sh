#!/bin/bash
# Dev (GPU box): where do the MFMA kernels lose their matrix-pipe time? Times layers with parts of the K loop compiled out
# (make -C openset-imagenet_amd/csrc ablate -> tools/probes/bin/libosi_hip_abl<bits>.so, -DOSI_ABLATE=<bits>: 1 no global loads in the
# loop, 2 no register-side staging / LDS stores, 4 no barriers, 8 no tap mask (all-taps weight gradient), 16 no LDS operand reads,
# 32 no epilogue (forward, input gradient)). Wrong results, right timing.   usage: tools/probes/ablate.sh [fwd|dgrad|wgrad|wgrad3|all]
cd "$(dirname "$0")/../.."
what=${1:-all}
run() {   # run <tool> <env assignment or ""> <variants...>
  tool=$1; e=$2; shift 2
  for round in 1 2; do for v in "$@"; do env $e OSI_DEV=1 OSI_HIP_LIB=$PWD/tools/probes/bin/libosi_hip_abl$v.so python tools/$tool 2>/dev/null || exit 1; done; done
}
if [ $what = fwd ] || [ $what = all ]; then
  echo "forward as the executor calls it (a = fused input activation), TFLOP/s at B = 128: 3x3 64@56a | 64->256@56a (row walker: not ablated) | 256->64@56 | 3x3 128@28a | 512->128@28 | 3x3 256@14a | 256->1024@14a | 1024->256@14 | 3x3 512@7a | 512->2048@7a"
  run time_fwd.py X=0 0 1 2 4 7 23 32 55
fi
if [ $what = dgrad ] || [ $what = all ]; then
  echo "input gradient, in-block fused form: 3x3 64@56 | 64->256@56 | 3x3 128@28 | 128->512@28 | 3x3 256@14 | 256->1024@14 | 3x3 512@7 | 512->2048@7"
  run time_dgrad.py X=0 0 1 2 4 7 23 32 55
fi
if [ $what = wgrad ] || [ $what = all ]; then
  echo "per-tap weight gradient, 1x1 layers (a = fused input activation): 256->64@56 | 64->256@56a | 512->128@28 | 128->512@28a | 1024->256@14 | 256->1024@14a | 2048->512@7 | 512->2048@7a"
  run time_wgrad.py OSI_TW_SHAPES=1x1 0 1 2 4 7 23
fi
if [ $what = wgrad3 ] || [ $what = all ]; then
  echo "all-taps weight gradient + fused activation: 3x3 64@56 | 128@28 | 256@14 | 512@7 | s2 128@56 | s2 256@28"
  run time_wgrad.py X=0 0 1 2 4 7 8 15 31
fi
