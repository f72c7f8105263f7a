"""Per-kernel statistics of the STEADY-STATE steps of a rocprofv3 --kernel-trace CSV
(MIOpen's solver search during warm-up launches hundreds of candidate kernels that would
otherwise dominate --stats).  Steps are delimited by the oks_nms kernel that ends each step.

    python tools/trace_stats.py <kernel_trace.csv> [skip_steps] > profiles/<name>.txt
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ends = [i for i, r in enumerate(rows) if 'oks_nms' in r['Kernel_Name']]
    if len(ends) <= skip:
        raise SystemExit('not enough steps in the trace')
    first, last = ends[skip - 1] + 1, ends[-1] + 1
    sel = rows[first:last]
    nsteps = len(ends) - skip
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg[r['Kernel_Name']]
        a[0] += d
        a[1] += 1
    busy = sum(a[0] for a in agg.values())
    wall = int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])
    print(f'# {path}: {nsteps} steady-state steps, {len(sel)} dispatches')
    print(f'# GPU busy {busy / nsteps / 1e6:.2f} ms/step, wall {wall / nsteps / 1e6:.2f} ms/step')
    print(f'# {"ms/step":>9} {"%":>6} {"calls/step":>10} {"avg us":>10}  kernel')
    for name, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
        print(f'{t / nsteps / 1e6:10.3f} {100 * t / busy:6.2f} {n / nsteps:10.1f} {t / n / 1e3:10.1f}  {name[:110]}')
    small = [(name, t, n) for name, (t, n) in agg.items() if t / n < 40e3]
    print(f'# launches shorter than 40 us: {sum(n for _, _, n in small) / nsteps:.0f} per step, '
          f'{sum(t for _, t, _ in small) / nsteps / 1e6:.2f} ms per step; by count:')
    for name, t, n in sorted(small, key=lambda x: -x[2])[:40]:
        print(f'{n / nsteps:10.1f} calls {t / n / 1e3:8.1f} us  {name[:130]}')


if __name__ == '__main__':
    main()
