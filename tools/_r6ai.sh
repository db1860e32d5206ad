cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/occ_pmc -o d -- python3 $R/tools/gated_bench.py config5 > /dev/null 2>&1
f=$(find $R/gpurun_out/occ_pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in rows:
    k=r['Kernel_Name'][:60]
    if 'spmm_csr_ordered' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVES': n[k]+=1
for k,c in acc.items():
    w=c['SQ_WAVES']
    print(k, 'launches', n[k], 'waves/launch', w/n[k])
    for name,v in c.items():
        if name!='SQ_WAVES': print('    ', name, round(v/w,1), 'per wave')
PY
rm -rf $R/gpurun_out/occ_pmc
