# round 6, batch c: ordered BPR backward -- tests, the stress table, the steady sports ranking call
cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py tests/test_gpu_round4.py "tests/test_gpu_parity.py" -q -m gpu -x 2>&1 | tail -15
echo "== stress: sharded"
for load in none busy; do
  echo "-- load=$load"; timeout 1500 python tools/stream_stress.py --model sharded --variants one_atomic,two_atomic,one,two,two_eager --trials 50 --load $load 2>&1 | grep RESULT | cut -c1-700
done
echo "== stress: unsharded, busy"; timeout 900 python tools/stream_stress.py --model unsharded --variants one_atomic,one,two --trials 50 --load busy 2>&1 | grep RESULT | cut -c1-700
echo "== steady sports ranking call, selection grid"
for g in 0 4096 8192 16384; do
  echo "-- CHAOREC_SEL_GRID=$g"; CHAOREC_SEL_GRID=$g EPOCH_APART=1 timeout 300 python tools/score_profile.py 3000 2>&1 | grep "epoch apart"
done
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_steady_r06c
EPOCH_APART=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o steady -- python3 $GRAFT_REPO_ROOT/tools/score_profile.py 3000 > $GRAFT_REPO_ROOT/gpurun_out/prof_steady_r06c.log 2>&1 < /dev/null
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r06_c_sports_steady_score_kernel_stats.csv; grep -i "score\|pack\|topk" "$f" | cut -c1-160; fi
