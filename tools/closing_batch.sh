# the closing batch of a round (TAG=r06_w bash tools/closing_batch.sh under gpurun): whole GPU suite, PMC traffic of the final spmm.hip, model profiles, the driver's bench command
cd $GRAFT_REPO_ROOT
echo "== full GPU suite"; timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "^  /\|Warning\|warnings.warn\|^$" | tail -15 | cut -c1-300
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== PMC traffic"; timeout 3000 python tools/collect_profiles.py ${TAG:-r06_w} sports config5 config5_full > gpurun_out/${TAG:-r06_w}_collect.log 2>&1; tail -5 gpurun_out/${TAG:-r06_w}_collect.log | cut -c1-200
cp gpurun_out/profiles_${TAG:-r06_w}/spmm_traffic_*.json profiles/ 2>/dev/null
echo "== model profiles"; timeout 1500 python tools/collect_model_profiles.py 2>&1 | tail -2 | cut -c1-300
cp gpurun_out/model_kernel_times.json profiles/model_kernel_times.json
echo "== bench"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG:-r06_w}_bench.out 2> gpurun_out/${TAG:-r06_w}_bench.err; tail -c 3300 gpurun_out/${TAG:-r06_w}_bench.out; cp bench_detail.json gpurun_out/${TAG:-r06_w}_bench_detail.json
echo; echo "== rocprofv3 --kernel-trace --stats of the same command"
cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG:-r06_w}_default_stats -o d -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG:-r06_w}_bench_under_rocprof.out 2>/dev/null
f=$(find $GRAFT_REPO_ROOT/gpurun_out/${TAG:-r06_w}_default_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${TAG:-r06_w}_default_kernel_stats.csv && head -8 "$f" | cut -c1-160
rm -rf $GRAFT_REPO_ROOT/gpurun_out/${TAG:-r06_w}_default_stats
