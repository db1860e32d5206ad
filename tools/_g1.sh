python -m pytest tests/test_gpu_round4.py tests/test_gpu_dist2.py -x -q -k "frontier or rowsparse_backward_kernels or sharded" 2>&1 | tail -4
CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 python bench.py --dataset config5_shard --dim 128 --steps 5 --warmup 2 --no-cpu-baseline --no-trained-state --no-hbm-regime --no-models > gpurun_out/r05_j_sharded_forced.json 2> gpurun_out/r05_j_err.log
python bench.py --dataset config5_shard --dim 128 --steps 5 --warmup 2 --no-cpu-baseline --no-trained-state --no-hbm-regime --spmm-only > gpurun_out/r05_j_unsharded.json 2>> gpurun_out/r05_j_err.log
