# the whole GPU suite once, everything kept: which tests fail inside the suite that pass alone, and how
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu 2>&1 > gpurun_out/suite_full.log
tail -5 gpurun_out/suite_full.log | cut -c1-300
grep -n "^FAILED\|^ERROR" gpurun_out/suite_full.log | head
grep -n "^E  " gpurun_out/suite_full.log | head -40 | cut -c1-400
