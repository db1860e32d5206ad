#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r03_m
export TMPDIR=/tmp
for m in MMGCN FREEDOM; do
  CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 timeout 600 python bench.py --model $m --gpus 1 --steps 50 --warmup 5 > ${O}_${m}_sharded1.json 2> ${O}_${m}_sharded1.err
  timeout 600 python bench.py --model $m --steps 50 --warmup 5 > ${O}_${m}_n1.json 2> ${O}_${m}_n1.err
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03_m_prof_$m" -o stats -- python3 "$GRAFT_REPO_ROOT/bench.py" --model $m --steps 300 --warmup 5 > /dev/null 2> "$GRAFT_REPO_ROOT/${O}_${m}_prof.err")
  f=$(find gpurun_out/r03_m_prof_$m -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" ${O}_${m}_kernel_stats.csv
  rm -rf gpurun_out/r03_m_prof_$m
done
for f in ${O}_*.json; do echo $f; tail -n 1 $f | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config']['model_class'], j['config']['launch'])"; done
python - <<'PY'
import csv,glob
for f in sorted(glob.glob('gpurun_out/r03_m_*_kernel_stats.csv')):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    glue=sum(float(r['TotalDurationNs']) for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name'] or 'rocprim' in r['Name'])
    print(f, 'total ms %.1f'%(tot/1e6), 'torch/rocclr glue share %.3f'%(glue/tot))
PY
