# End-of-round record (through gpurun from the repository root; <commit> = the build): the GPU suite, smoke(), the
# rocprofv3 block of the 1-degree grid, the driver's bench command with its wall time.
mkdir -p gpurun_out/r4z
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4z/full.log 2>&1 || { tail -30 gpurun_out/r4z/full.log; exit 1; }
tail -2 gpurun_out/r4z/full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4z/smoke.log 2>&1 || { tail -20 gpurun_out/r4z/smoke.log; exit 1; }
tail -2 gpurun_out/r4z/smoke.log
bash profiles/profile_r04.sh g1 ${1:-unknown} > gpurun_out/r4z/prof.log 2>&1 || { tail -20 gpurun_out/r4z/prof.log; exit 1; }
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r4z/default.json 2> gpurun_out/r4z/default.err ) 2> gpurun_out/r4z/default.time
tail -3 gpurun_out/r4z/default.time
head -6 gpurun_out/prof_r04/summary/global_one_degree_streamed_kernel_stats.csv | cut -c1-140
