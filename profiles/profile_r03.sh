#!/bin/bash
# Round-3 profiling recipe (run on the GPU box through gpurun from the repository root):
#     bash profiles/profile_r03.sh <block|all> <commit>
# kernel-trace stats and separate PMC passes (never combined with other trace domains), as
# MI355X_MICROARCH.md prescribes; the program itself directly after `--`.  Raw output under
# gpurun_out/prof_r03/<tag>/, summaries (what gets committed) under gpurun_out/prof_r03/summary/ ->
# copy to profiles/r03/.  A pass that fails removes its block from the summary: nothing stale is
# ever summarised (every pass starts from an empty directory).  <commit> = the build that is profiled
# (`git rev-parse --short HEAD` where the repository is; the GPU box has no .git).
set -o pipefail
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r03
SUM=$OUT/summary
which=${1:-all}
export GRAVHMC_PROFILED_COMMIT=${2:-unknown}
mkdir -p $SUM
cd /tmp
prof() {  # tag, rocprof options..., --, bench options...   (returns non-zero if the pass failed)
  local tag=$1; shift
  local ropts=()
  while [ "$1" != "--" ]; do ropts+=("$1"); shift; done
  shift
  rm -rf $OUT/$tag $OUT/$tag.json $OUT/$tag.err
  if ! timeout -k 10 400 rocprofv3 "${ropts[@]}" --output-format csv -d $OUT/$tag -- python3 $REPO/bench.py "$@" > $OUT/$tag.json 2> $OUT/$tag.err; then
    echo "$tag FAILED"; tail -5 $OUT/$tag.err; rm -rf $OUT/$tag; return 1
  fi
  echo "$tag done"
}
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU"
SQ2="SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
if [ $which = all ] || [ $which = c4 ]; then
  # one chain, matrix-free: the team pass (default) -- and the column-per-workgroup pass for comparison
  A="--workload c4_global_tesseroid --matrix-free --no-cpu-baseline"
  if prof c4mf_trace --kernel-trace --stats -- $A --steps 200 --warmup 20 \
     && prof c4mf_sq --pmc $SQ1 --kernel-trace -- $A --steps 20 --warmup 0 \
     && prof c4mf_sq2 --pmc $SQ2 --kernel-trace -- $A --steps 20 --warmup 0; then
    python3 $REPO/profiles/summarize.py stats $OUT/c4mf_trace $SUM/c4_matrix_free_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/c4_matrix_free_pmc_summary.json "rocprofv3 --pmc passes (SQ counters, with --kernel-trace only) of python3 bench.py $A --steps 20 --warmup 0 on MI355X; recipe profiles/profile_r03.sh; per-dispatch averages; SQ cycle counters in quad-cycles summed over the chip" $OUT/c4mf_sq $OUT/c4mf_sq2
    cp $OUT/c4mf_trace.json $SUM/bench_c4_matrix_free_under_rocprof_trace.json
  fi
  export GRAVHMC_MF_TEAM=0
  if prof c4mf0_trace --kernel-trace --stats -- $A --steps 200 --warmup 20; then
    python3 $REPO/profiles/summarize.py stats $OUT/c4mf0_trace $SUM/c4_matrix_free_column_per_workgroup_kernel_stats.csv
    cp $OUT/c4mf0_trace.json $SUM/bench_c4_matrix_free_column_per_workgroup_under_rocprof_trace.json
  fi
  unset GRAVHMC_MF_TEAM
fi
if [ $which = all ] || [ $which = c4b ]; then
  A="--workload c4_global_tesseroid --matrix-free --chains-per-gpu 8 --no-cpu-baseline"
  if prof c4b_trace --kernel-trace --stats -- $A --steps 100 --warmup 20 \
     && prof c4b_sq --pmc $SQ1 --kernel-trace -- $A --steps 20 --warmup 0 \
     && prof c4b_sq2 --pmc $SQ2 --kernel-trace -- $A --steps 20 --warmup 0; then
    python3 $REPO/profiles/summarize.py stats $OUT/c4b_trace $SUM/c4_matrix_free_8chains_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/c4_matrix_free_8chains_pmc_summary.json "rocprofv3 --pmc passes (SQ counters, with --kernel-trace only) of python3 bench.py $A --steps 20 --warmup 0 on MI355X; recipe profiles/profile_r03.sh; per-dispatch averages; SQ cycle counters in quad-cycles summed over the chip" $OUT/c4b_sq $OUT/c4b_sq2
    cp $OUT/c4b_trace.json $SUM/bench_c4_matrix_free_8chains_under_rocprof_trace.json
  fi
fi
if [ $which = all ] || [ $which = c4b2 ]; then
  # the two-pass form of the same workload (no co-residency needed): GRAVHMC_MFB_FUSED=0
  export GRAVHMC_MFB_FUSED=0
  A="--workload c4_global_tesseroid --matrix-free --chains-per-gpu 8 --no-cpu-baseline"
  if prof c4b2_trace --kernel-trace --stats -- $A --steps 100 --warmup 20 \
     && prof c4b2_sq --pmc $SQ1 --kernel-trace -- $A --steps 20 --warmup 0 \
     && prof c4b2_sq2 --pmc $SQ2 --kernel-trace -- $A --steps 20 --warmup 0; then
    python3 $REPO/profiles/summarize.py stats $OUT/c4b2_trace $SUM/c4_matrix_free_8chains_two_pass_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/c4_matrix_free_8chains_two_pass_pmc_summary.json "GRAVHMC_MFB_FUSED=0 rocprofv3 --pmc passes (SQ counters, with --kernel-trace only) of python3 bench.py $A --steps 20 --warmup 0 on MI355X; recipe profiles/profile_r03.sh; per-dispatch averages; SQ cycle counters in quad-cycles summed over the chip" $OUT/c4b2_sq $OUT/c4b2_sq2
    cp $OUT/c4b2_trace.json $SUM/bench_c4_matrix_free_8chains_two_pass_under_rocprof_trace.json
  fi
  unset GRAVHMC_MFB_FUSED
fi
if [ $which = all ] || [ $which = c4s ]; then
  A="--workload c4_global_tesseroid --shift-invariant --no-cpu-baseline"
  if prof c4s_trace --kernel-trace --stats -- $A --steps 2000 --warmup 200 \
     && prof c4s_sq --pmc $SQ1 --kernel-trace -- $A --steps 200 --warmup 0 \
     && prof c4s_sq2 --pmc $SQ2 --kernel-trace -- $A --steps 200 --warmup 0; then
    python3 $REPO/profiles/summarize.py stats $OUT/c4s_trace $SUM/c4_shift_invariant_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/c4_shift_invariant_pmc_summary.json "rocprofv3 --pmc passes (SQ counters, with --kernel-trace only) of python3 bench.py $A --steps 200 --warmup 0 on MI355X; recipe profiles/profile_r03.sh; per-dispatch averages" $OUT/c4s_sq $OUT/c4s_sq2
    cp $OUT/c4s_trace.json $SUM/bench_c4_shift_invariant_under_rocprof_trace.json
  fi
fi
if [ $which = all ] || [ $which = c2 ]; then
  A="--no-cpu-baseline --no-extra"
  if prof c2_trace --kernel-trace --stats -- $A --steps 30 --warmup 10 \
     && prof c2_fetch --pmc FETCH_SIZE --kernel-trace -- $A --steps 10 --warmup 0 \
     && prof c2_write --pmc WRITE_SIZE --kernel-trace -- $A --steps 10 --warmup 0; then
    python3 $REPO/profiles/summarize.py stats $OUT/c2_trace $SUM/c2_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/c2_pmc_summary.json "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: one per pass, with --kernel-trace only) of python3 bench.py --steps 10 --warmup 0 $A (workload C2) on MI355X; recipe profiles/profile_r03.sh; per-dispatch averages, sizes in KB as rocprofv3 reports them" $OUT/c2_fetch $OUT/c2_write
    cp $OUT/c2_trace.json $SUM/bench_c2_under_rocprof_trace.json
  fi
fi
if [ $which = all ] || [ $which = c2b ]; then
  # C2 with 16 chains per GPU on the stored kernel: teams reading G once per step, and the two-pass batch
  A="--workload c2_uniform_100x100x50 --chains-per-gpu 16 --no-cpu-baseline --no-extra --batch-team on"
  if prof c2b_trace --kernel-trace --stats -- $A --steps 40 --warmup 10 \
     && prof c2b_sq --pmc $SQ1 --kernel-trace -- $A --steps 10 --warmup 0 \
     && prof c2b_sq2 --pmc $SQ2 --kernel-trace -- $A --steps 10 --warmup 0 \
     && prof c2b_fetch --pmc FETCH_SIZE --kernel-trace -- $A --steps 10 --warmup 0; then
    python3 $REPO/profiles/summarize.py stats $OUT/c2b_trace $SUM/c2_16chains_one_read_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/c2_16chains_one_read_pmc_summary.json "rocprofv3 --pmc passes (SQ counters; FETCH_SIZE in its own pass; with --kernel-trace only) of python3 bench.py $A --steps 10 --warmup 0 on MI355X; recipe profiles/profile_r03.sh; per-dispatch averages; SQ cycle counters in quad-cycles summed over the chip; FETCH_SIZE in KB as rocprofv3 reports it" $OUT/c2b_sq $OUT/c2b_sq2 $OUT/c2b_fetch
    cp $OUT/c2b_trace.json $SUM/bench_c2_16chains_one_read_under_rocprof_trace.json
  fi
  A="--workload c2_uniform_100x100x50 --chains-per-gpu 16 --no-cpu-baseline --no-extra --batch-team off"
  if prof c2b2_trace --kernel-trace --stats -- $A --steps 40 --warmup 10; then
    python3 $REPO/profiles/summarize.py stats $OUT/c2b2_trace $SUM/c2_16chains_two_passes_kernel_stats.csv
    cp $OUT/c2b2_trace.json $SUM/bench_c2_16chains_two_passes_under_rocprof_trace.json
  fi
fi
if [ $which = all ] || [ $which = c1 ]; then
  A="--workload c1_uniform_20x30x10 --no-cpu-baseline"
  if prof c1_trace --kernel-trace --stats -- $A --steps 20000 --warmup 2000 \
     && prof c1_sq --pmc $SQ1 --kernel-trace -- $A --steps 20000 --warmup 0 \
     && prof c1_sq2 --pmc $SQ2 --kernel-trace -- $A --steps 20000 --warmup 0; then
    python3 $REPO/profiles/summarize.py stats $OUT/c1_trace $SUM/c1_resident_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/c1_resident_pmc_summary.json "rocprofv3 --pmc passes (SQ counters, with --kernel-trace only) of python3 bench.py $A --steps 20000 --warmup 0 on MI355X; recipe profiles/profile_r03.sh; per-dispatch averages" $OUT/c1_sq $OUT/c1_sq2
    cp $OUT/c1_trace.json $SUM/bench_c1_resident_under_rocprof_trace.json
  fi
fi
cd $REPO
ls -la $SUM
