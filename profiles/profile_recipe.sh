#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel-trace stats of the default bench
# and two separate PMC passes (FETCH_SIZE / WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
set -o pipefail
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}   # copy the summaries you keep into profiles/rNN/
mkdir -p $OUT
REPO=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace -- python3 $REPO/bench.py --steps 30 --warmup 10 --no-cpu-baseline > $REPO/$OUT/bench_trace.json 2> $REPO/$OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc_fetch -- python3 $REPO/bench.py --steps 10 --warmup 0 --no-cpu-baseline > $REPO/$OUT/bench_pmc_fetch.json 2> $REPO/$OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc_write -- python3 $REPO/bench.py --steps 10 --warmup 0 --no-cpu-baseline > $REPO/$OUT/bench_pmc_write.json 2> $REPO/$OUT/pmc_write.err
cd $REPO
find $OUT -name "*.csv" | head -20
