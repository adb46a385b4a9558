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
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
cd /tmp && export TMPDIR=/tmp
echo "[1/5] bench with cpu_baseline"
timeout -k 10 500 python3 $R/bench.py --steps 5 --warmup 2 > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "[2/5] kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof.log 2>&1
echo "[3/5] pmc FETCH_SIZE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
echo "[4/5] pmc WRITE_SIZE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
echo "[5/5] pmc SQ"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o run --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
# the per-dispatch traces are large; keep what the summaries need
find $OUT/prof -name "*kernel_trace.csv" -delete
tail -1 $OUT/bench_full.json
echo "[6] reconstruction kernels at the bench shape"
rm -rf $OUT/prof_recon
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_recon -o run --output-format csv -- python3 $R/tools/reconstruct_scale.py 50000 > $OUT/prof_recon.log 2>&1
find $OUT/prof_recon -name "*kernel_trace.csv" -delete
grep "reconstruct:" $OUT/prof_recon.log | tail -1
