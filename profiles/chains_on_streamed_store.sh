# Several chains per GPU on the streamed harmonic store (1-degree global grid): every chain a light context on the shared
# tables, its four launches per step on its own stream (gh_batch_*, csrc/gravhmc.hip kids_run).  Through gpurun from the
# repository root; prints chains, chain-steps/s, ms per lock-step of the batch.
mkdir -p gpurun_out/r4s
for c in 2 4 8; do
  python bench.py --workload x3_global_one_degree --shift-invariant --chains-per-gpu $c --no-cpu-baseline --no-extra --steps 400 --warmup 40 > gpurun_out/r4s/g1_c$c.json 2> gpurun_out/r4s/g1_c$c.err
done
python - <<'P'
import json
for c in (2, 4, 8):
    try:
        d = json.loads([l for l in open("gpurun_out/r4s/g1_c%d.json" % c) if l.startswith("{")][-1])
        print(c, round(d["value"], 1), d["ms_per_step"], d["config"].get("accepted"), d["config"].get("trajectories"))
    except Exception as e:
        print(c, "ERR", e, open("gpurun_out/r4s/g1_c%d.err" % c).read()[-600:])
P
