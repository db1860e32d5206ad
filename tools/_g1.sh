timeout 900 python -m pytest tests/test_gpu_round5.py -q -m gpu 2>&1 | tail -3
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_zz_ranges -o s -- python3 $GRAFT_REPO_ROOT/tools/score_ranges_bench.py 300000 2000000 128 > $GRAFT_REPO_ROOT/gpurun_out/r05_zz_ranges.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/prof_stats.py gpurun_out/r05_zz_ranges 12 | grep -i 'class\|pack\|sweep'
