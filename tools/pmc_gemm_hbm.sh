#!/bin/bash
# HBM traffic of the split GEMM at three shapes (FETCH_SIZE / WRITE_SIZE in separate --pmc passes):
# is A re-read from HBM by the column tiles of a row tile, or served by the XCD's L2?
# usage (GPU box): tools/pmc_gemm_hbm.sh <outdir under gpurun_out>
OUT=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
i=0
for shape in "625044 256 1024" "625044 1024 256" "625044 256 256"; do
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
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
    rd = sum(sorted(d['FETCH_SIZE'])[1:-1]) / max(1, len(d['FETCH_SIZE']) - 2) * 1024 * 2 / 1e9 if d.get('FETCH_SIZE') else float('nan')
    wr = sum(sorted(d['WRITE_SIZE'])[1:-1]) / max(1, len(d['WRITE_SIZE']) - 2) * 1024 / 1e9 if d.get('WRITE_SIZE') else float('nan')
    print(f'grid {grid}: HBM read {rd:.2f} GB (FETCH_SIZE x 2, gfx950 16-B/lane correction), written {wr:.2f} GB per launch')
PY
