cd $GRAFT_REPO_ROOT
run() { echo "== flags: $1"; CHAOREC_EXTRA_HIPCC_FLAGS="$1" timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-regime --no-full-config5 --no-models 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['config'].get('gene_ranklist_ms'))"; }
run ""
run "-DCHAOREC_SPMM_MINW=4"
run "-DCHAOREC_SPMM_MINW=5"
run "-DCHAOREC_SPMM_MINW=6"
run "-DCHAOREC_SPMM_MINW=8"
run ""
