#!/bin/bash
# MFMA-pipe utilisation of the dense kernels on the bench.py workload: one --pmc pass (SQ + GRBM
# counters only, no trace domains), summarised per kernel.
# usage: tools/pmc_bench_mfma.sh <outdir under gpurun_out> [bench args]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_]*MFMA[A-Z_0-9]*" | sort -u > $R/gpurun_out/$OUT/mfma_counters_available.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv \
  -d $R/gpurun_out/$OUT/p1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-native-side --no-secondary --no-extra-passes "$@" > $R/gpurun_out/$OUT/p1.log 2>&1
python3 - <<PY
import csv, glob, collections, re
rows = []
for f in glob.glob('$R/gpurun_out/$OUT/p1/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Dispatch_Id']))
ends = sorted({int(r['Dispatch_Id']) for r in rows if 'oks_nms' in r['Kernel_Name']})
lo, hi = ends[-2], ends[-1]                       # the last (steady-state) step
disp = collections.defaultdict(dict)
for r in rows:
    d = int(r['Dispatch_Id'])
    if lo < d <= hi:
        disp[d]['k'] = r['Kernel_Name']
        disp[d][r['Counter_Name']] = float(r['Counter_Value'])
def cls(k):
    if k.startswith('Cijk'):
        return 'hipBLASLt fp32 GEMM ' + re.search(r'MT\\d+x\\d+x\\d+', k).group(0)
    if 'conv_nhwc_kernel' in k:
        return 'pave conv_nhwc_kernel (Bottleneck tail, fp32 MFMA)'
    if 'gemm_bf16x3' in k or 'gemm_q_' in k or 'gemm_w' in k or 'gemm_s_' in k or 'bottleneck_chain' in k or 'stem7x7_q' in k:
        return 'pave gemm_q / gemm_w / gemm_wn / gemm_q_ln / gemm_s / bottleneck_chain / stem7x7_q kernels (split GEMM / convolutions, bf16 MFMA x 6)'
    if 'enc_tile_kernel' in k:
        return 'pave enc_tile_kernel (encoder sampling, no MFMA)'
    if 'grouped_conv_fwd' in k or k.startswith('igemm_fwd'):
        return 'MIOpen / CK fp32 convolution'
    if 'fused_deform_attn' in k:
        return 'pave fused_deform_attn_kernel (sampling, no MFMA)'
    return 'other (elementwise, norms, copies, top-k ...)'
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for d in disp.values():
    if 'GRBM_GUI_ACTIVE' not in d:
        continue
    a = agg[cls(d['k'])]
    a[0] += d['GRBM_GUI_ACTIVE'] / 8.0            # summed over the 8 XCDs -> kernel cycles
    a[1] += d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    a[2] += 1
tot = sum(a[0] for a in agg.values())
tb = sum(a[1] for a in agg.values())
with open('$R/gpurun_out/$OUT/mfma_util.txt', 'w') as fo:
    fo.write('# MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), last step of\n')
    fo.write('# bench.py under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE\n')
    fo.write('# %step_cycles  mfma_util  dispatches  kernel class\n')
    for k, (gui, busy, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        fo.write(f'{100 * gui / tot:8.2f}  {busy / (gui * 1024.0) if gui else 0:8.3f}  {n:6d}  {k}\n')
    fo.write(f'# whole step: MFMA pipe busy {tb / (tot * 1024.0):.3f} of all SIMD cycles\n')
print(open('$R/gpurun_out/$OUT/mfma_util.txt').read())
PY
rm -rf $R/gpurun_out/$OUT/p1
