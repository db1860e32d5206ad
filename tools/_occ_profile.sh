# kernel-trace evidence for the gated launch's occupancy settings: tools/gated_bench.py under rocprofv3, product build and -DCHAOREC_SPMM_SP_HIOCC=0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in product hiocc0; do
  if [ $v = hiocc0 ]; then export CHAOREC_EXTRA_HIPCC_FLAGS="-DCHAOREC_SPMM_SP_HIOCC=0"; else unset CHAOREC_EXTRA_HIPCC_FLAGS; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/occ_$v -o d -- python3 $R/tools/gated_bench.py config5 > $R/gpurun_out/occ_$v.out 2>/dev/null
  f=$(find $R/gpurun_out/occ_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -v "amdgpu.ids\|Warn" $R/gpurun_out/occ_$v.out | tail -3
  head -1 "$f"; grep "spmm_csr_ordered_kernel" "$f" | cut -c1-60,200-400
  cp "$f" $R/gpurun_out/occ_${v}_kernel_stats.csv; rm -rf $R/gpurun_out/occ_$v
done
