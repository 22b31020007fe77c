# The streamed sweep held to the registers of 5 / 6 / 8 waves per SIMD (GRAVHMC_LW_OCC) against the default (4): 1-degree grid.
mkdir -p gpurun_out/r4t
B="python bench.py --workload x3_global_one_degree --shift-invariant --no-cpu-baseline --no-extra --steps 1000 --warmup 100"
$B > gpurun_out/r4t/occ0.json 2> gpurun_out/r4t/err.log
for k in 5 6 8; do GRAVHMC_LW_OCC=$k $B > gpurun_out/r4t/occ$k.json 2>> gpurun_out/r4t/err.log; done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4t/*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], round(d['value'], 1), round(d['roofline']['avg_ms'] * 1e3, 1), 'us/pass', d['config'].get('accepted'), d['config']['final_U'][0])
    except Exception as e:
        print(f, 'ERR', e)
P
