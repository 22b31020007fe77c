set -o pipefail
export TMPDIR=/tmp
REPO=$(pwd)
O=$REPO/gpurun_out/r4m
mkdir -p $O
A="--workload c4_global_tesseroid --shift-invariant --steps 3000 --warmup 300 --no-extra --no-cpu-baseline"
for v in 0 1; do
  GRAVHMC_LONSYM_FUSED=$v timeout -k 10 300 python3 bench.py $A > $O/c4_f$v.json 2> $O/c4_f$v.err || exit 1
  python3 -c "import json;l=json.loads(open('$O/c4_f$v.json').read().strip().splitlines()[-1]);print('fused=$v',l['value'])"
done
cd /tmp
rm -rf $O/c4_f1_trace
GRAVHMC_LONSYM_FUSED=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_f1_trace -- python3 $REPO/bench.py $A > $O/c4_f1_trace.json 2> $O/c4_f1_trace.err || exit 1
f=$(find $O/c4_f1_trace -name '*kernel_stats.csv' | head -1)
cut -c1-160 $f | head -8
