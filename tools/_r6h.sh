# round 6, batch h: the whole GPU suite, the driver's bench command, the models' one-stream profiles, smoke
cd $GRAFT_REPO_ROOT
echo "== full GPU suite"; timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== profiles"; timeout 1500 python tools/collect_model_profiles.py 2>&1 | tail -3
mkdir -p profiles && cp gpurun_out/model_kernel_times.json profiles/model_kernel_times.json
echo "== bench"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_h_bench.out 2> gpurun_out/r06_h_bench.err; tail -c 3600 gpurun_out/r06_h_bench.out; cp bench_detail.json gpurun_out/r06_h_bench_detail.json
