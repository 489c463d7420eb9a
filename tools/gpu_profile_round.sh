#!/bin/bash
# Everything profiles/rNN_* is made of, in one call on the GPU box (results under gpurun_out/$1/, copy them to profiles/):
#   gpurun --timeout 3000 -- 'bash tools/gpu_profile_round.sh r03p r03'
# PMC passes (FETCH_SIZE, WRITE_SIZE; two separate passes each, never combined with other traces) + tools/pmc_summary.py for the
# 640x480 stream, the 1241x376 stream and the exact mode; the kernel-trace statistics of the default (guarded fast) and the exact
# mode; the bench lines.  Run it after the LAST edit of ur-mvo_amd/csrc: bench.py refuses a PMC summary whose source_sha differs
# from the kernel sources.
OUT=gpurun_out/${1:-prof}; TAG=${2:-rXX}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
pmc() {   # $1 = suffix of the summary file, rest = extra bench arguments
  SUF=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace -d $R/$OUT/pmc_$c -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-exact-check --no-guard-calibration "$@" > $R/$OUT/pmc_$c$SUF.log 2>&1
  done
  (cd $R && python tools/pmc_summary.py $(find $OUT/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/pmc_WRITE_SIZE -name "*.db" | head -1) $OUT/${TAG}_pmc_hbm$SUF.json "python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-exact-check --no-guard-calibration $*")
  rm -rf $R/$OUT/pmc_FETCH_SIZE $R/$OUT/pmc_WRITE_SIZE
  cp $R/$OUT/${TAG}_pmc_hbm$SUF.json $R/profiles/ 2>/dev/null    # so that the bench lines below carry roofline.traffic
}
pmc ""
pmc _1241x376 --resolution 1241x376
pmc _exact --precision 0
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_fast -o st -- python3 $R/bench.py --steps 15 --warmup 2 --repeats 1 --no-cpu-baseline --no-exact-check --no-guard-calibration > $R/$OUT/stats_fast.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_exact -o st -- python3 $R/bench.py --precision 0 --steps 15 --warmup 2 --repeats 1 --no-cpu-baseline > $R/$OUT/stats_exact.log 2>&1
cd $R
cp $(find $OUT/stats_fast -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_fast_kernel_stats.csv
cp $(find $OUT/stats_exact -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_exact_kernel_stats.csv
rm -rf $OUT/stats_fast $OUT/stats_exact
timeout 300 python bench.py > $OUT/${TAG}_bench_fast_640x480.json 2> $OUT/bench.err
timeout 300 python bench.py --resolution 1241x376 --no-cpu-baseline > $OUT/${TAG}_bench_fast_1241x376.json 2>> $OUT/bench.err
timeout 300 python bench.py --precision 0 --no-cpu-baseline > $OUT/${TAG}_bench_exact_640x480.json 2>> $OUT/bench.err
timeout 300 python bench.py --precision 1 --no-cpu-baseline > $OUT/${TAG}_bench_unguarded_640x480.json 2>> $OUT/bench.err
timeout 300 python bench.py --precision 1 --resolution 1241x376 --no-cpu-baseline > $OUT/${TAG}_bench_unguarded_1241x376.json 2>> $OUT/bench.err
ls -la $OUT
