set -e
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
rm -rf $OUT/prof $OUT/prof_mammals
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 python3 $R/bench.py --steps 5 --warmup 2 > $OUT/bench_full.json 2> $OUT/bench_full.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
timeout -k 10 200 python3 $R/tools/mammals_calls.py 300 > $OUT/mammals_calls.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_mammals -o run --output-format csv -- python3 $R/tools/mammals_calls.py 100 > $OUT/prof_mammals.log 2>&1
find $OUT/prof $OUT/prof_mammals -name "*kernel_trace.csv" -delete
cd $R
tail -1 $OUT/bench_full.json | cut -c1-300
grep -v amdgpu $OUT/mammals_calls.log
