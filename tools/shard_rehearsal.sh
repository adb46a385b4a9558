#!/bin/bash
# Run on the GPU box from the repo root: what each rank of an N-way run would do, one shard after the other on the one GPU
# (bench.py --emulate-shard r/N: the library's shard plan, no collective) -- first under the predicted plan, then under the
# plan rebalanced by the times just measured (what the ranks of an N > 1 run do during set-up), and once more.
#   gpurun --timeout 1100 -- 'bash tools/shard_rehearsal.sh 8'
set -e
N=${1:-8}
run_pass() {   # $1: output file, $2: extra arguments
  : > $1
  for r in $(seq 0 $((N-1))); do
    timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --emulate-shard $r/$N $2 >> $1 2>> gpurun_out/shards_$N.err
  done
  python3 - <<PY
import json
rows=[json.loads(l) for l in open("$1")]
ms=[r["ms_per_step"] for r in rows]
print("ms per shard:", [round(x,2) for x in ms], "largest over mean %.3f" % (max(ms)/(sum(ms)/len(ms))), "K2 TF/s", [round(r["roofline"]["achieved"],1) for r in rows])
open("gpurun_out/.shard_times","w").write(",".join("%.4f" % x for x in ms))
PY
}
echo "predicted plan"
run_pass gpurun_out/shards_${N}_predicted.jsonl ""
echo "rebalanced once from those times"
T1=$(cat gpurun_out/.shard_times)
run_pass gpurun_out/shards_${N}_rebalanced1.jsonl "--shard-times $T1"
echo "rebalanced a second time (--rebalance-steps 2)"
run_pass gpurun_out/shards_$N.jsonl "--shard-times $T1;$(cat gpurun_out/.shard_times)"
