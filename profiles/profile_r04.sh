#!/bin/bash
# Round-4 profiling recipe (run on the GPU box through gpurun from the repository root):
#     bash profiles/profile_r04.sh <block|all> <commit>
# kernel-trace stats and separate PMC passes (never combined with other trace domains), as
# MI355X_MICROARCH.md prescribes; the program itself directly after `--`.  Raw output under
# gpurun_out/prof_r04/<tag>/, summaries (what gets committed) under gpurun_out/prof_r04/summary/ ->
# copy to profiles/r04/.  A pass that fails removes its block from the summary (every pass starts from an
# empty directory).  <commit> = the build that is profiled (the GPU box has no .git).
set -o pipefail
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r04
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
NOTE="on MI355X; recipe profiles/profile_r04.sh; per-dispatch averages; SQ cycle counters in quad-cycles summed over the chip"
if [ $which = all ] || [ $which = c2 ]; then
  # the headline: the driver's own command (all extra runs are children: only the parent's kernels are traced)
  A="--steps 20 --warmup 5 --no-extra --no-cpu-baseline"
  if prof c2_trace --kernel-trace --stats -- $A \
     && prof c2_fetch --pmc FETCH_SIZE --kernel-trace -- $A \
     && prof c2_write --pmc WRITE_SIZE --kernel-trace -- $A; then
    python3 $REPO/profiles/summarize.py stats $OUT/c2_trace $SUM/c2_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/c2_pmc_summary.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate, with --kernel-trace only) of python3 bench.py $A $NOTE" $OUT/c2_fetch $OUT/c2_write
    cp $OUT/c2_trace.json $SUM/bench_c2_under_rocprof_trace.json
  fi
fi
if [ $which = all ] || [ $which = c1b ]; then
  # C1 (north_star's target configuration), 16 chains in lock-step inside resident_batch_kernel
  A="--workload c1_uniform_20x30x10 --chains-per-gpu 16 --no-cpu-baseline --no-extra"
  if prof c1b_trace --kernel-trace --stats -- $A --steps 4000 --warmup 400 \
     && prof c1b_sq --pmc $SQ1 --kernel-trace -- $A --steps 1000 --warmup 0 \
     && prof c1b_sq2 --pmc $SQ2 --kernel-trace -- $A --steps 1000 --warmup 0 \
     && prof c1b_fetch --pmc FETCH_SIZE --kernel-trace -- $A --steps 1000 --warmup 0 \
     && prof c1b_write --pmc WRITE_SIZE --kernel-trace -- $A --steps 1000 --warmup 0; then
    python3 $REPO/profiles/summarize.py stats $OUT/c1b_trace $SUM/c1_16chains_lockstep_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/c1_16chains_lockstep_pmc_summary.json "rocprofv3 --pmc passes (SQ counters, FETCH_SIZE, WRITE_SIZE: separate passes, with --kernel-trace only) of python3 bench.py $A --steps 1000 --warmup 0 $NOTE" $OUT/c1b_sq $OUT/c1b_sq2 $OUT/c1b_fetch $OUT/c1b_write
    cp $OUT/c1b_trace.json $SUM/bench_c1_16chains_lockstep_under_rocprof_trace.json
  fi
  A3="--workload c3_segment_wavelet3d_tv --chains-per-gpu 16 --no-cpu-baseline --no-extra"
  if prof c3b_trace --kernel-trace --stats -- $A3 --steps 4000 --warmup 400; then
    python3 $REPO/profiles/summarize.py stats $OUT/c3b_trace $SUM/c3_16chains_lockstep_kernel_stats.csv
    cp $OUT/c3b_trace.json $SUM/bench_c3_16chains_lockstep_under_rocprof_trace.json
  fi
fi
if [ $which = all ] || [ $which = c4s ]; then
  # C4 on the shift-invariant store, harmonic domain, one launch per phase (GRAVHMC_LONSYM_RESIDENT=0: what runs with
  # a stencil regulariser or when the persistent launch gives up): one chain, and 8 chains on the shared tables
  A="--workload c4_global_tesseroid --shift-invariant --no-cpu-baseline --no-extra"
  export GRAVHMC_LONSYM_RESIDENT=0
  if prof c4s_trace --kernel-trace --stats -- $A --steps 2000 --warmup 200 \
     && prof c4s_sq --pmc $SQ1 --kernel-trace -- $A --steps 200 --warmup 0 \
     && prof c4s_fetch --pmc FETCH_SIZE --kernel-trace -- $A --steps 200 --warmup 0; then
    python3 $REPO/profiles/summarize.py stats $OUT/c4s_trace $SUM/c4_shift_invariant_harmonic_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/c4_shift_invariant_harmonic_pmc_summary.json "rocprofv3 --pmc passes (SQ counters; FETCH_SIZE in its own pass; with --kernel-trace only) of python3 bench.py $A --steps 200 --warmup 0 $NOTE" $OUT/c4s_sq $OUT/c4s_fetch
    cp $OUT/c4s_trace.json $SUM/bench_c4_shift_invariant_harmonic_under_rocprof_trace.json
  fi
  unset GRAVHMC_LONSYM_RESIDENT
  if prof c4s8_trace --kernel-trace --stats -- $A --chains-per-gpu 8 --steps 2000 --warmup 200; then
    python3 $REPO/profiles/summarize.py stats $OUT/c4s8_trace $SUM/c4_shift_invariant_8chains_kernel_stats.csv
    cp $OUT/c4s8_trace.json $SUM/bench_c4_shift_invariant_8chains_under_rocprof_trace.json
  fi
fi
if [ $which = all ] || [ $which = c4p ]; then
  # C4, one chain (BASELINE configs[3]: one chain per GPU): the harmonic pass as ONE persistent launch per batch of
  # trajectories (lonsymh_resident_kernel, csrc/lonres.hip.h)
  A="--workload c4_global_tesseroid --shift-invariant --no-cpu-baseline --no-extra"
  if prof c4p_trace --kernel-trace --stats -- $A --steps 8000 --warmup 800 \
     && prof c4p_sq --pmc $SQ1 --kernel-trace -- $A --steps 2000 --warmup 0 \
     && prof c4p_fetch --pmc FETCH_SIZE --kernel-trace -- $A --steps 2000 --warmup 0 \
     && prof c4p_write --pmc WRITE_SIZE --kernel-trace -- $A --steps 2000 --warmup 0; then
    python3 $REPO/profiles/summarize.py stats $OUT/c4p_trace $SUM/c4_shift_invariant_persistent_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/c4_shift_invariant_persistent_pmc_summary.json "rocprofv3 --pmc passes (SQ counters; FETCH_SIZE, WRITE_SIZE in passes of their own; with --kernel-trace only) of python3 bench.py $A --steps 2000 --warmup 0 $NOTE" $OUT/c4p_sq $OUT/c4p_fetch $OUT/c4p_write
    cp $OUT/c4p_trace.json $SUM/bench_c4_shift_invariant_persistent_under_rocprof_trace.json
  fi
fi
if [ $which = all ] || [ $which = c5r ]; then
  # the 96 GB row block one of 8 GPUs holds of C5 (row-block sharding, world = 1: the collective is a local copy)
  A="--workload c5_uniform_200x200x60 --rows-fraction 8 --shard --shard-axis rows --no-cpu-baseline --no-extra"
  if prof c5r_trace --kernel-trace --stats -- $A --steps 20 --warmup 5; then
    python3 $REPO/profiles/summarize.py stats $OUT/c5r_trace $SUM/c5_row_block_share_kernel_stats.csv
    cp $OUT/c5r_trace.json $SUM/bench_c5_row_block_share_under_rocprof_trace.json
  fi
fi
if [ $which = all ] || [ $which = g1 ]; then
  # a 1-degree global grid (648 000 cells x 65 341 observations) on the STREAMED harmonic store (csrc/lonsymw.hip.h)
  A="--workload x3_global_one_degree --shift-invariant --no-cpu-baseline --no-extra"
  if prof g1_trace --kernel-trace --stats -- $A --steps 2000 --warmup 200 \
     && prof g1_sq --pmc $SQ1 --kernel-trace -- $A --steps 400 --warmup 0 \
     && prof g1_fetch --pmc FETCH_SIZE --kernel-trace -- $A --steps 400 --warmup 0; then
    python3 $REPO/profiles/summarize.py stats $OUT/g1_trace $SUM/global_one_degree_streamed_kernel_stats.csv
    python3 $REPO/profiles/summarize.py pmc $SUM/global_one_degree_streamed_pmc_summary.json "rocprofv3 --pmc passes (SQ counters; FETCH_SIZE in its own pass; with --kernel-trace only) of python3 bench.py $A --steps 400 --warmup 0 $NOTE" $OUT/g1_sq $OUT/g1_fetch
    cp $OUT/g1_trace.json $SUM/bench_global_one_degree_streamed_under_rocprof_trace.json
  fi
fi
ls -la $SUM
