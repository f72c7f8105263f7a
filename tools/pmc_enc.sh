#!/bin/bash
# Collects PMC counters for the sampling kernels (separate passes; no trace domains mixed in).
# usage: tools/pmc_enc.sh <outdir> [bench_kernels args]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr TD_BUSY_avr TA_TA_BUSY_sum" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$OUT/p$i -- python3 $R/tools/bench_kernels.py "$@" > $R/gpurun_out/$OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$R/gpurun_out/$OUT/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:84]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'deform' in k or 'msda' in k or 'enc_' in k:
        print(k)
        for c, v in sorted(d.items()):
            print(f'   {c:36s} n={len(v):4d} mean={sum(v)/len(v):.4g}')
PY
