#!/bin/bash
# Run on the GPU box from the repo root: what each rank of an N-way run would do, one shard after the other on the one GPU
# (bench.py --emulate-shard r/N: the library's shard plan, no collective), plus a 2-rank gloo rehearsal of the launcher path.
#   gpurun --timeout 1100 -- 'bash tools/shard_rehearsal.sh 8'
set -e
N=${1:-8}
OUT=gpurun_out/shards_$N.jsonl
: > $OUT
for r in $(seq 0 $((N-1))); do
  timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --emulate-shard $r/$N >> $OUT 2>> gpurun_out/shards_$N.err
  echo "shard $r/$N done"
done
python3 - <<PY
import json
rows=[json.loads(l) for l in open("$OUT")]
ms=[r["ms_per_step"] for r in rows]
print("ms per shard:", [round(x,2) for x in ms], "spread %.3f" % (max(ms)/min(ms)), "K2 TF/s", [round(r["roofline"]["achieved"],1) for r in rows])
PY
