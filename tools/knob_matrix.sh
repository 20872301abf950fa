#!/bin/bash
# Dev tool (GPU box): the whole-network parity tests under every alternate execution path the executor keeps (A/B switches), so that a
# path which is not the default cannot rot unnoticed. Prints one line per configuration.
set -o pipefail
T="tests/test_gate_pinned_gpu.py tests/test_model_gpu.py::test_forward_backward_vs_oracle_and_golden tests/test_model_gpu.py::test_training_run_is_bitwise_reproducible tests/test_model_gpu.py::test_backward_schedules_are_bitwise_equivalent tests/test_model_gpu.py::test_eval_mode_forward tests/test_eval_fused_gpu.py::test_fused_inference_forward_equals_the_training_topology"
CFGS=("OSI_NO_OVERLAP=1" "OSI_TAIL_SPLIT=0" "OSI_STEM_DIRECT=0" "OSI_STEM_FUSED=0" "OSI_STEM_POOL_STATS=0" "OSI_WGRAD3=0" "OSI_WGRAD3=1" "OSI_BN_WIDE_P=0" "OSI_BN_SINGLE_P=1" "OSI_FWD_FORK=0" "OSI_TAIL_CUS=64" "OSI_DS_SPARSE=0" "OSI_STEM_WGRAD_MAIN=0" "OSI_DGRAD_WIDE=1" "OSI_DP_RESERVED_CUS=8" "OSI_FWD_ROWS=0" "OSI_FWD_ROWS=2" "OSI_WGRAD_TILE=64" "OSI_FWD_W3=0" "OSI_DGRAD_W3=0" "OSI_FWD_WINO=0" "OSI_DGRAD_WINO=0" "OSI_FWD_WINO=0 OSI_DGRAD_WINO=0" "OSI_WINO_STREAMK=0" "OSI_WINO_STREAMK=1" "OSI_WINO_STREAMK=3" "OSI_WGRAD_WINO=0" "OSI_FWD_WINO=0 OSI_DGRAD_WINO=0 OSI_WGRAD_WINO=0" "OSI_WINO_WIDE=0" "OSI_WINO_WEIGHTS_ASIDE=0" "OSI_FWD_WIDE=1" "OSI_FWD_WINO=0 OSI_FWD_W3=0" "OSI_EVAL_FUSED=0")
# optional arguments: run only these configurations (one gpurun call holds ~20 of them)
if [ $# -gt 0 ]; then CFGS=("$@"); fi
for cfg in "${CFGS[@]}"; do
  out=$(env $cfg timeout -k 10 300 python -m pytest $T -x -q -m gpu -k "not benchmarked_batch" 2>&1 | tail -1)
  echo "$cfg : $out"
done
