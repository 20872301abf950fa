#!/bin/bash
# Dev tool (GPU box, via gpurun): the end-of-round record on ONE box — the default bench line, the other workloads, the batch sweep.
#   tools/final_numbers.sh r05    ->  gpurun_out/<tag>_bench_final.json, <tag>_other_workloads.txt, <tag>_batch_sweep.txt
TAG=${1:-r05}; O=gpurun_out
python bench.py > $O/${TAG}_bench_final.json 2> $O/${TAG}_bench_final.err || exit 1
echo "default line done" >&2
: > $O/${TAG}_other_workloads.txt
for w in p1 p3; do
  python bench.py --workload $w --no-cpu-baseline --sustained-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['workload'])" >> $O/${TAG}_other_workloads.txt
done
python bench.py --force-dp --no-cpu-baseline --sustained-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('rccl', {}); print('force-dp', d['value'], d['ms_per_step'], r.get('per_bucket_comm_ms'), r.get('exposed_comm_ms'), r.get('channels'))" >> $O/${TAG}_other_workloads.txt
echo "other workloads done" >&2
bash tools/batch_sweep.sh > $O/${TAG}_batch_sweep.txt
echo "batch sweep done" >&2
