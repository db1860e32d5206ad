export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_zz_mmgcn -o s -- python3 $GRAFT_REPO_ROOT/bench.py --model MMGCN --steps 100 --warmup 10 > $GRAFT_REPO_ROOT/gpurun_out/r05_zz_mmgcn.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05_zz_mmgcn.err
cd $GRAFT_REPO_ROOT
python3 tools/prof_stats.py gpurun_out/r05_zz_mmgcn 28
tail -c 300 gpurun_out/r05_zz_mmgcn.json
