#!/bin/bash
# Dev tool: what the PRODUCT Winograd kernel (k_wino: forward / fused input gradient) pays for, part by part — -DOSI_FABL=<bits> builds of
# conv_wino.hip (WRONG results by design; bits in conv_wino.hip: 2 = no patch loads in the K loop, 4 = no patch transform / LDS stores,
# 16 = one output quad of four in the epilogue, 31 = matrix work + loop skeleton only), timed per layer shape by tools/time_wino.py.
#   here:        tools/ab_wino_ablation.sh build          -> .ab/libosi_fabl<bits>.so (hipcc cross-compiles; .ab/ travels with gpurun)
#   on the box:  tools/ab_wino_ablation.sh run [B]        -> gpurun_out/fabl/out.txt       (profiles/r06_wino_product_ablation.txt)
ROOT=$(cd "$(dirname "$0")/.." && pwd); CSRC=$ROOT/openset-imagenet_amd/csrc; BITS="2 4 6 16 31"
if [ "$1" = build ]; then
  mkdir -p "$ROOT/.ab"
  objs=$(ls "$CSRC"/*.o | grep -v "conv_wino.o\|stamps\|diag")
  for f in $BITS; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize -DOSI_FABL=$f \
        -c "$CSRC/conv_wino.hip" -o "$ROOT/.ab/conv_wino_f$f.o" \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/.ab/libosi_fabl$f.so" "$ROOT/.ab/conv_wino_f$f.o" $objs ) &
  done
  wait; ls -la "$ROOT"/.ab/libosi_fabl*.so
else
  B=${2:-128}; mkdir -p "$ROOT/gpurun_out/fabl"; : > "$ROOT/gpurun_out/fabl/out.txt"
  for f in 0 $BITS; do
    if [ $f = 0 ]; then L=$CSRC/libosi_hip.so; else L=$ROOT/.ab/libosi_fabl$f.so; fi
    echo "== OSI_FABL $f" >> "$ROOT/gpurun_out/fabl/out.txt"
    OSI_DEV=1 OSI_HIP_LIB=$L timeout -k 10 120 python "$ROOT/tools/time_wino.py" $B wino 2>/dev/null | grep -v "^B=" >> "$ROOT/gpurun_out/fabl/out.txt" || exit 1
  done
  cat "$ROOT/gpurun_out/fabl/out.txt"
fi
