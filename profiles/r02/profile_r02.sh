#!/bin/bash
# Round-2 profiling recipe (run on the GPU box through gpurun from the repository root):
#   kernel-trace stats + separate PMC passes (never combined with other trace domains), as
#   MI355X_MICROARCH.md prescribes.  Raw output under gpurun_out/prof_r02/, summaries (what is
#   committed) under gpurun_out/prof_r02/summary/ -> copy to profiles/r02/.
set -o pipefail
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r02
SUM=$OUT/summary
mkdir -p $SUM
cd /tmp
prof() {  # tag, rocprof options..., --, bench options...
  local tag=$1; shift
  local ropts=()
  while [ "$1" != "--" ]; do ropts+=("$1"); shift; done
  shift
  rocprofv3 "${ropts[@]}" --output-format csv -d $OUT/$tag -- python3 $REPO/bench.py "$@" > $OUT/$tag.json 2> $OUT/$tag.err || { echo "$tag FAILED"; tail -5 $OUT/$tag.err; }
  echo "$tag done"
}
which=${1:-all}
if [ $which = all ] || [ $which = c2 ]; then
  prof c2_trace --kernel-trace --stats -- --steps 30 --warmup 10 --no-cpu-baseline --no-extra
  prof c2_fetch --pmc FETCH_SIZE --kernel-trace -- --steps 10 --warmup 0 --no-cpu-baseline --no-extra
  prof c2_write --pmc WRITE_SIZE --kernel-trace -- --steps 10 --warmup 0 --no-cpu-baseline --no-extra
  python3 $REPO/profiles/summarize.py stats $OUT/c2_trace $SUM/c2_kernel_stats.csv
  python3 $REPO/profiles/summarize.py pmc $SUM/c2_pmc_summary.json "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: one per pass, with --kernel-trace only) of python3 bench.py --steps 10 --warmup 0 --no-cpu-baseline --no-extra (workload C2) on MI355X; recipe profiles/profile_r02.sh; per-dispatch averages, sizes in KB as rocprofv3 reports them" $OUT/c2_fetch $OUT/c2_write
  cp $OUT/c2_trace.json $SUM/bench_c2_under_rocprof_trace.json
fi
if [ $which = all ] || [ $which = c4 ]; then
  prof c4mf_trace --kernel-trace --stats -- --workload c4_global_tesseroid --matrix-free --steps 60 --warmup 10 --no-cpu-baseline
  prof c4mf_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --kernel-trace -- --workload c4_global_tesseroid --matrix-free --steps 20 --warmup 0 --no-cpu-baseline
  prof c4mf_sq2 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVES_EQ_64 SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE --kernel-trace -- --workload c4_global_tesseroid --matrix-free --steps 20 --warmup 0 --no-cpu-baseline
  python3 $REPO/profiles/summarize.py stats $OUT/c4mf_trace $SUM/c4_matrix_free_kernel_stats.csv
  python3 $REPO/profiles/summarize.py pmc $SUM/c4_matrix_free_pmc_summary.json "rocprofv3 --pmc passes (SQ counters, with --kernel-trace only) of python3 bench.py --workload c4_global_tesseroid --matrix-free --steps 20 --warmup 0 --no-cpu-baseline on MI355X; recipe profiles/profile_r02.sh; per-dispatch averages; SQ cycle counters in quad-cycles summed over the chip" $OUT/c4mf_sq $OUT/c4mf_sq2
  cp $OUT/c4mf_trace.json $SUM/bench_c4_matrix_free_under_rocprof_trace.json
fi
if [ $which = all ] || [ $which = c5 ]; then
  prof c5_trace --kernel-trace --stats -- --workload c5_uniform_200x200x60 --cells-fraction 8 --steps 40 --warmup 10 --no-cpu-baseline
  prof c5_fetch --pmc FETCH_SIZE --kernel-trace -- --workload c5_uniform_200x200x60 --cells-fraction 8 --steps 20 --warmup 0 --no-cpu-baseline
  python3 $REPO/profiles/summarize.py stats $OUT/c5_trace $SUM/c5_share_kernel_stats.csv
  python3 $REPO/profiles/summarize.py pmc $SUM/c5_share_pmc_summary.json "rocprofv3 --pmc FETCH_SIZE (with --kernel-trace only) of python3 bench.py --workload c5_uniform_200x200x60 --cells-fraction 8 --steps 20 --warmup 0 --no-cpu-baseline (the 96 GB share one of 8 GPUs holds of C5) on MI355X; recipe profiles/profile_r02.sh" $OUT/c5_fetch
  cp $OUT/c5_trace.json $SUM/bench_c5_share_under_rocprof_trace.json
fi
if [ $which = all ] || [ $which = c1 ]; then
  prof c1_trace --kernel-trace --stats -- --workload c1_uniform_20x30x10 --steps 20000 --warmup 2000 --no-cpu-baseline
  python3 $REPO/profiles/summarize.py stats $OUT/c1_trace $SUM/c1_resident_kernel_stats.csv
  cp $OUT/c1_trace.json $SUM/bench_c1_resident_under_rocprof_trace.json
fi
cd $REPO
ls -la $SUM
