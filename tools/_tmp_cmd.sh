export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round4.py -q -m gpu -x -k "rowlist or light" > gpurun_out/r04_x_tests.log 2>&1; echo rc=$?; tail -3 gpurun_out/r04_x_tests.log
for T in 64 128 512; do
(cd /tmp && CHAOREC_ROWLIST_LONG_T=$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_x_prof_T$T -o shard -- python3 $R/bench.py --dataset config5_shard --dim 128 --steps 10 --warmup 3 --no-hbm-regime --no-cpu-baseline --no-trained-state > $R/gpurun_out/r04_x_T$T.json 2> $R/gpurun_out/r04_x_T$T.err); echo T=$T rc=$?
done
