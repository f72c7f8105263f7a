#!/bin/bash
# SQ / TA / TCP counters of the encoder tile kernel (28 frames per launch, bench offset sizes):
# what bounds it -- VALU issue, LDS, the texture path, waiting?   usage: tools/pmc_enc_tile.sh <outdir>
OUT=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$OUT/p$i -- python3 $R/tools/bench_kernels.py --frames 28 --sigma 0.9 --enc-only --prepared > $R/gpurun_out/$OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in sorted(glob.glob('$R/gpurun_out/$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'enc_tile_kernel<14' in r['Kernel_Name'] and ', true>' in r['Kernel_Name']:   # the prepared-mode instantiation
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m = {}
for c, v in sorted(agg.items()):
    v = sorted(v)[len(v) // 4: len(v) - len(v) // 4] or v
    m[c] = sum(v) / len(v)
    print(f'{c:32s} {m[c]:16.0f}')
if 'GRBM_GUI_ACTIVE' in m:
    cyc = m['GRBM_GUI_ACTIVE'] / 8
    print(f'# kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs): {cyc:.0f}')
    if 'SQ_WAVE_CYCLES' in m:
        wc = m['SQ_WAVE_CYCLES']
        for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS',
                  'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VMEM'):
            if k in m:
                print(f'# {k} / SQ_WAVE_CYCLES = {m[k] / wc:.3f}')
        print(f'# resident waves per SIMD = SQ_WAVE_CYCLES x 4 / (1024 SIMDs x cycles) = {wc * 4 / (1024 * cyc):.2f}')
    if 'SQ_LDS_IDX_ACTIVE' in m:
        print(f'# LDS array busy = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles) = {m["SQ_LDS_IDX_ACTIVE"] / (256 * cyc):.3f}; '
              f'bank-conflict share = {m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]:.3f}')
    if 'SQ_ACTIVE_INST_VALU' in m:
        print(f'# VALU issue busy per SIMD = SQ_ACTIVE_INST_VALU x 4 / (1024 x cycles) = {m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc):.3f}')
PY
