#!/bin/bash
# One gpurun call's worth of measurements (arguments: a tag, then the names of the parts to run).
#   tools/gpu_batch.sh r04_c tests models sharded gpus2 freedom_prof
tag=$1; shift
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
for part in "$@"; do
  case $part in
    tests)
      python -m pytest tests -x -q -m gpu > $out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/${tag}_tests.log ;;
    tests_new)
      python -m pytest tests/test_gpu_round4.py tests/test_gpu_sparse_family.py tests/test_gpu_dist2.py -q -m gpu > $out/${tag}_tests_new.log 2>&1; echo "tests_new rc=$?"; tail -3 $out/${tag}_tests_new.log ;;
    bench)
      python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?" ;;
    models)
      for m in FREEDOM MMGCN; do
        python bench.py --model $m --steps 50 --warmup 10 > $out/${tag}_bench_$m.json 2> $out/${tag}_bench_$m.err; echo "$m rc=$?"
        CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 python bench.py --model $m --steps 50 --warmup 10 > $out/${tag}_bench_${m}_sharded.json 2> $out/${tag}_bench_${m}_sharded.err; echo "$m sharded rc=$?"
      done ;;
    sharded)
      CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_LightGCN_sharded.json 2> $out/${tag}_bench_LightGCN_sharded.err; echo "sharded rc=$?"; tail -c 400 $out/${tag}_bench_LightGCN_sharded.err ;;
    gpus2)
      python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_gpus2.json 2> $out/${tag}_bench_gpus2.err; echo "gpus2 rc=$?"; tail -c 400 $out/${tag}_bench_gpus2.err ;;
    freedom_prof)
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_prof_FREEDOM -o FREEDOM -- python3 $GRAFT_REPO_ROOT/bench.py --model FREEDOM --steps 100 --warmup 10 > $GRAFT_REPO_ROOT/$out/${tag}_prof_FREEDOM.json 2> $GRAFT_REPO_ROOT/$out/${tag}_prof_FREEDOM.err); echo "freedom_prof rc=$?"
      python3 tools/prof_stats.py $out/${tag}_prof_FREEDOM 14 ;;
    mmgcn_prof)
      (cd /tmp && CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_prof_MMGCN_sharded -o MMGCN -- python3 $GRAFT_REPO_ROOT/bench.py --model MMGCN --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/$out/${tag}_prof_MMGCN_sharded.json 2> $GRAFT_REPO_ROOT/$out/${tag}_prof_MMGCN_sharded.err); echo "mmgcn_prof rc=$?" ;;
    mmgcn_sharded)
      CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 python bench.py --model MMGCN --steps 50 --warmup 10 > $out/${tag}_bench_MMGCN_sharded.json 2> $out/${tag}_bench_MMGCN_sharded.err; echo "MMGCN sharded (two streams) rc=$?"
      CHAOREC_DIST_MMGCN_STREAMS=0 CHAOREC_FORCE_SHARDED=1 CHAOREC_FORCE_COLLECTIVES=1 python bench.py --model MMGCN --steps 50 --warmup 10 > $out/${tag}_bench_MMGCN_sharded_1stream.json 2> $out/${tag}_bench_MMGCN_sharded_1stream.err; echo "MMGCN sharded (one stream) rc=$?" ;;
    overlap)
      python tools/score_overlap_exp.py > $out/${tag}_score_overlap_ub3.txt 2>&1; echo "overlap ub3 rc=$?"; cat $out/${tag}_score_overlap_ub3.txt | tail -6
      CHAOREC_EXTRA_HIPCC_FLAGS="-DCHAOREC_PF_UB64=2" python tools/score_overlap_exp.py > $out/${tag}_score_overlap_ub2.txt 2>&1; echo "overlap ub2 rc=$?"; cat $out/${tag}_score_overlap_ub2.txt | tail -6 ;;
    rccl_streams)
      python tools/rccl_streams_repro.py > $out/${tag}_rccl_streams_repro.txt 2>&1; echo "rccl_streams rc=$?"; cat $out/${tag}_rccl_streams_repro.txt ;;
    *) echo "unknown part $part" ;;
  esac
done
