#!/bin/bash
# Run on the GPU box from the repo root: what the L2 misses of K2's two input streams cost.  Libraries built with
#   -D'CAFE_EXPERIMENT_A_SLOT(s)=0' -D'CAFE_EXPERIMENT_A_ROW(r)=0'   (a0: every workgroup stages rows of ONE matrix at row tile 0)
#   -D'CAFE_EXPERIMENT_B_COLUMN(c)=0'                                (b0: every workgroup stages column tile 0 of the child panel)
# (prune_gemm.hip; wrong results, the same work) are expected under cafexp_amd/_k2v/lib_{a0,b0,a0b0}.so.
mkdir -p gpurun_out
cp cafexp_amd/libcafe_mi355x.so /tmp/lib_base.so
: > gpurun_out/k2_l2_ab.log
run() {
  timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
p = d['phases_ms_per_step']
print('%-6s ms_per_step %.3f  K2 %.3f  rest of prune %.3f' % ('$1', d['ms_per_step'], p['prune_gemm'], p['prune_total'] - p['prune_gemm']))
" >> gpurun_out/k2_l2_ab.log 2>&1
}
for rep in 1 2; do
  run base
  for v in a0 b0 a0b0; do cp cafexp_amd/_k2v/lib_$v.so cafexp_amd/libcafe_mi355x.so; run $v; done
  cp /tmp/lib_base.so cafexp_amd/libcafe_mi355x.so
done
cat gpurun_out/k2_l2_ab.log
