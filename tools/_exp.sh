set -e
ROOT=$PWD
mkdir -p gpurun_out/exp
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "adamw or trajectory or rccl or loader" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/exp/st3 -- python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $ROOT/gpurun_out/exp/st.log 2>&1
python3 - <<PY
import csv,glob,os
f=sorted(glob.glob("$ROOT/gpurun_out/exp/st3/*/*_kernel_stats.csv"), key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:40]:
    if 'adamw' in r['Name'] or 'multi_tensor' in r['Name']: print(r['Name'].replace('dmet::(anonymous namespace)::','')[:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
cd $ROOT
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
