#!/bin/bash
# HBM-side traffic of the encoder sampling kernel on the bench.py workload itself: FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (no trace domains mixed in), averaged per launch.
# usage: tools/pmc_bench_enc.sh <outdir under gpurun_out>
OUT=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$OUT/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-native-side --no-secondary --no-extra-passes > $R/gpurun_out/$OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json, hashlib
_src = open('$R/pavenet_amd/csrc/pave_enc_tile.hip', 'rb').read()
blob = hashlib.sha1(b'blob %d\0' % len(_src) + _src).hexdigest()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$R/gpurun_out/$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:120]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in agg.items():
    if 'enc_tile_kernel' in k or 'enc_head_major_kernel' in k:
        kname = 'enc_tile_kernel' if 'enc_tile_kernel' in k else 'enc_head_major_kernel'
        for c, v in sorted(d.items()):
            v = sorted(v)[len(v) // 4: len(v) - len(v) // 4] or v   # inter-quartile mean
            out[c] = dict(n=len(d[c]), mean=sum(v) / len(v))
            print(k, c, out[c])
# rocprofv3 reports both counters in KiB-like units of the guide's HBM section: FETCH_SIZE /
# WRITE_SIZE are in kilobytes; gfx950 tallies 128-B read requests of 16-B/lane loads at 64 B ->
# reads are doubled (MI355X_MICROARCH.md, HBM).
if 'FETCH_SIZE' in out and 'WRITE_SIZE' in out:
    rd = out['FETCH_SIZE']['mean'] * 1024 * 2
    wr = out['WRITE_SIZE']['mean'] * 1024
    res = dict(kernel=kname + ' (encoder MSDA, T=1)', workload='bench.py default (28 frames/launch)', frames_per_launch=28,
               fetch_size_kb_raw=out['FETCH_SIZE']['mean'], write_size_kb=out['WRITE_SIZE']['mean'],
               read_bytes_corrected=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr, kernel_source_git_blob=blob,
               note='FETCH_SIZE doubled per the gfx950 correction for 16-B/lane loads; separate --pmc passes')
    json.dump(res, open('$R/gpurun_out/$OUT/enc_kernel_traffic.json', 'w'), indent=1)
    print(json.dumps(res))
PY
