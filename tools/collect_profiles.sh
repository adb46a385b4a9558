#!/bin/bash
# Run on the GPU box from the repo root: collects everything profiles/ is summarised from.
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh'
# then here:  python tools/summarize_profiles.py ...   (see that file)
# Separate passes (kernel trace first, then one --pmc set per pass) as MI355X_MICROARCH.md prescribes;
# the profiled program is python3 itself (no env/bash hop between rocprofv3 and the process that opens the GPU).
set -e
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/prof_mammals
cd /tmp && export TMPDIR=/tmp
echo "[1/8] bench with cpu_baseline, parity sample, one column per family, other configs"
timeout -k 10 600 python3 $R/bench.py --steps 5 --warmup 2 > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "[2/8] kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
echo "[3/8] pmc FETCH_SIZE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
echo "[4/8] pmc WRITE_SIZE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
echo "[5/8] pmc SQ"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_sq.log 2>&1
echo "[6/8] mammals calls (BASELINE configs 2, 3)"
timeout -k 10 200 python3 $R/tools/mammals_calls.py 300 > $OUT/mammals_calls.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_mammals -o run --output-format csv -- python3 $R/tools/mammals_calls.py 100 > $OUT/prof_mammals.log 2>&1
# the per-dispatch traces are large; keep what the summaries need
find $OUT/prof $OUT/prof_mammals -name "*kernel_trace.csv" -delete
cd $R
tail -1 $OUT/bench_full.json | cut -c1-300
echo "[7/8] eight shards, one after the other (what each rank of an 8-GPU run does)"
bash tools/shard_rehearsal.sh 8 | tail -4
echo "[8/8] the launcher path: python bench.py --gpus 2 (gloo rehearsal, both ranks on this GPU) + per-launch table + shard kernel trace"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 3 --no-cpu-baseline > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err
cut -c1-200 $OUT/bench_gloo2.json
timeout -k 10 200 python3 tools/launch_table.py > $OUT/launch_table_full.txt 2>/dev/null
timeout -k 10 200 python3 tools/launch_table.py 3/8 > $OUT/launch_table_shard3.txt 2>/dev/null
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_shard -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --emulate-shard 3/8 > $OUT/prof_shard.log 2>&1
find $OUT/prof_shard -name "*kernel_trace.csv" -delete
echo "[9] second-tier paths at the bench shape: reconstruction (kernel trace), p-values"
rm -rf $OUT/prof_recon
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_recon -o run --output-format csv -- python3 $R/tools/reconstruct_scale.py > $OUT/prof_recon.log 2>&1
find $OUT/prof_recon -name "*kernel_trace.csv" -delete
cd $R
timeout -k 10 300 python3 tools/reconstruct_scale.py 2>/dev/null | grep -v amdgpu > $OUT/reconstruct_scale.log
timeout -k 10 300 python3 tools/pvalues_scale.py 2>/dev/null | grep -v amdgpu > $OUT/pvalues_scale.log || true
cd $R
