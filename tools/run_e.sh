#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r03_e
(timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > ${O}_gpu_tests.log 2>&1
for m in MMGCN FREEDOM; do
  timeout 600 python bench.py --model $m --steps 50 --warmup 5 > ${O}_${m}_n1.json 2> ${O}_${m}_n1.err
  CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 timeout 600 python bench.py --model $m --gpus 1 --steps 50 --warmup 5 > ${O}_${m}_sharded1.json 2> ${O}_${m}_sharded1.err
done
tail -n 6 ${O}_gpu_tests.log
for f in ${O}_*_n1.json ${O}_*_sharded1.json; do echo $f; tail -n 1 $f | cut -c1-600; done
tail -n 3 ${O}_*.err | cut -c1-300
