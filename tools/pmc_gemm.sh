#!/bin/bash
# SQ counters of the split GEMM kernel at three shapes of the model (FFN1, FFN2, layer4 conv1):
# MFMA-busy share, wait / issue-stall / active split of the wave cycles, LDS conflicts.
# usage (GPU box): tools/pmc_gemm.sh <outdir under gpurun_out>
OUT=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
i=0
for shape in "625044 256 1024" "625044 1024 256" "117600 1024 512"; do
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$OUT/p$i -- python3 $R/tools/gemm_one.py $shape > $R/gpurun_out/$OUT/p$i.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$R/gpurun_out/$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if any(t in r['Kernel_Name'] for t in ('gemm_bf16x3', 'gemm_q_', 'gemm_w')):
            agg[r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
for grid, d in agg.items():
    print('grid', grid)
    for c, v in sorted(d.items()):
        v = sorted(v)[len(v) // 4: len(v) - len(v) // 4] or v
        print(f'   {c:28s} {sum(v) / len(v):16.0f}')
PY
