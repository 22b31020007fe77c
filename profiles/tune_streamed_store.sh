# Tuning runs of the streamed harmonic store (csrc/lonsymw.hip.h) on the 1-degree global grid: parity tests first, then
# bench.py with the diagnostic switches (GRAVHMC_LW_BREAK: phases off, timing only).  Run through gpurun from the repository root.
mkdir -p gpurun_out/r4o
timeout -k 10 600 python -m pytest tests/test_gpu_mfbatch.py tests/test_gpu_oracle_pins.py -m gpu -x -q -s -k "shift_invariant or one_degree" > gpurun_out/r4o/t.log 2>&1 || { tail -30 gpurun_out/r4o/t.log; exit 1; }
tail -2 gpurun_out/r4o/t.log
B="python bench.py --workload x3_global_one_degree --shift-invariant --no-cpu-baseline --no-extra --steps 1000 --warmup 100"
$B > gpurun_out/r4o/g1.json 2> gpurun_out/r4o/g1.err
for k in 1 2 4 7; do GRAVHMC_LW_BREAK=$k $B > gpurun_out/r4o/g1_brk$k.json 2>> gpurun_out/r4o/g1.err; done
for w in 32 128; do GRAVHMC_LW_WAVES_PER_CU=$w $B > gpurun_out/r4o/g1_w$w.json 2>> gpurun_out/r4o/g1.err; done
for k in 16 32 56; do GRAVHMC_LW_LDS_PAD=$k $B > gpurun_out/r4o/g1_pad$k.json 2>> gpurun_out/r4o/g1.err; done
for k in 1 2 3 4; do GRAVHMC_LW_FWD=$k $B > gpurun_out/r4o/g1_fwd$k.json 2>> gpurun_out/r4o/g1.err; done
GRAVHMC_LW_FWD=2 GRAVHMC_LW_WAVES_PER_CU=32 $B > gpurun_out/r4o/g1_fwd2_w32.json 2>> gpurun_out/r4o/g1.err
GRAVHMC_LW_FWD=3 GRAVHMC_LW_WAVES_PER_CU=32 $B > gpurun_out/r4o/g1_fwd3_w32.json 2>> gpurun_out/r4o/g1.err
GRAVHMC_LONSYM_RESIDENT=0 python bench.py --workload c4_global_tesseroid --shift-invariant --no-cpu-baseline --no-extra --steps 4000 --warmup 400 > gpurun_out/r4o/c4_lpp.json 2>> gpurun_out/r4o/g1.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4o/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],1), round(d['roofline']['avg_ms']*1e3,1), 'us/pass', 'accepted', d['config'].get('accepted'), 'of', d['config'].get('trajectories'))
    except Exception as e: print(f, 'ERR', e)
P
