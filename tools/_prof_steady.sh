# per-kernel split of the STEADY sports ranking call (thresholds carried from one epoch earlier): rocprofv3 kernel stats of tools/score_profile.py
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_steady
EPOCH_APART=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o steady -- python3 $GRAFT_REPO_ROOT/tools/score_profile.py 3000 > $GRAFT_REPO_ROOT/gpurun_out/prof_steady.log 2>&1 < /dev/null
grep "epoch apart" $GRAFT_REPO_ROOT/gpurun_out/prof_steady.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r05_zzz_sports_steady_score_kernel_stats.csv; grep -i "score\|pack\|topk" "$f" | cut -c1-200; else echo "no kernel_stats.csv"; fi
