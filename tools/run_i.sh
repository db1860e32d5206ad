#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r03_i
timeout 900 python bench.py > ${O}_bench_line.json 2> ${O}_bench_line.err
timeout 900 python bench.py --steps 20 --warmup 5 > ${O}_bench_line_driver_args.json 2> ${O}_bench_line_driver_args.err
timeout 1500 python tools/collect_profiles.py r03_k sports config5 config5_full > ${O}_collect.log 2>&1
CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > ${O}_sharded_fused.json 2> ${O}_sharded_fused.err
(timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_real_data.py tests/test_gpu_fused_step.py tests/test_gpu_models.py -m gpu -q -x 2>&1 | tail -5) > ${O}_tests.log 2>&1
tail -c 400 ${O}_bench_line.err; tail -n 12 ${O}_collect.log | cut -c1-220; tail -n 3 ${O}_tests.log
ls gpurun_out/profiles_r03_k/
