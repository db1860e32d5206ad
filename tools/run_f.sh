#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r03_f
timeout 120 tools/micro/grid_barrier > ${O}_grid_barrier.txt 2>&1
for m in MMGCN FREEDOM; do
  CHAOREC_FORCE_SHARDED=1 timeout 600 python bench.py --model $m --gpus 1 --steps 50 --warmup 5 > ${O}_${m}_sharded1_nocoll.json 2> ${O}_${m}_sharded1_nocoll.err
  CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 timeout 600 python bench.py --model $m --gpus 1 --steps 50 --warmup 5 > ${O}_${m}_sharded1.json 2> ${O}_${m}_sharded1.err
done
CHAOREC_FORCE_SHARDED=1 timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > ${O}_lightgcn_sharded1_nocoll.json 2> ${O}_lightgcn_sharded1_nocoll.err
CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 CHAOREC_DIST_EXCHANGE=direct timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > ${O}_lightgcn_sharded1_direct.json 2> ${O}_lightgcn_sharded1_direct.err
CHAOREC_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline > ${O}_lightgcn_gloo2.json 2> ${O}_lightgcn_gloo2.err
cat ${O}_grid_barrier.txt
for f in ${O}_*.json; do python - "$f" <<'PY'
import json,sys
raw=open(sys.argv[1]).read()
l=[x for x in raw.splitlines() if x.startswith('{')]
if not l: print(sys.argv[1], 'NO JSON', raw[-300:]); sys.exit()
j=json.loads(l[-1]); c=j['config']
print(sys.argv[1].split('/')[-1], 'ms/step %.4f'%j['ms_per_step'], c.get('model_class',''), c['launch'][:60], '|', c['parallelism'][:110], '| rank', c.get('gene_ranklist_ms_incl_d2h_wall', c.get('gene_ranklist_ms')))
PY
done
tail -n 3 ${O}_lightgcn_gloo2.err | cut -c1-300
