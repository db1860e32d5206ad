#!/bin/bash
# GPU run B of round 3: block-joint selection + fused sharded step -- tests, smoke, bench, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r03_b
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_models.py tests/test_gpu_fused_step.py -m gpu -q -x -k "score or prefilter or rank or round3 or vbpr or fused or rows_mean or sharded" 2>&1 | tail -40) > ${O}_tests.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > ${O}_smoke.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-hbm-regime > ${O}_bench.json 2> ${O}_bench.err
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03_b_prof" -o stats -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-hbm-regime --steps 60 --warmup 5 > "$GRAFT_REPO_ROOT/${O}_prof_bench.json" 2> "$GRAFT_REPO_ROOT/${O}_prof.err")
f=$(find gpurun_out/r03_b_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" ${O}_kernel_stats.csv && head -25 "$f" | cut -c1-200 > ${O}_kernel_stats_head.txt
rm -rf gpurun_out/r03_b_prof
CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > ${O}_sharded_fused.json 2> ${O}_sharded_fused.err
CHAOREC_DIST_STEP=autograd CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > ${O}_sharded_autograd.json 2> ${O}_sharded_autograd.err
tail -n 8 ${O}_tests.log; cat ${O}_smoke.log; tail -c 300 ${O}_bench.err; tail -c 300 ${O}_sharded_fused.err
