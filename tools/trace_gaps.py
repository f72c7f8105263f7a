"""Idle gaps of the GPU inside the LAST steady-state step of a rocprofv3 --kernel-trace CSV (steps are
delimited by the oks_nms kernel): every gap >= min_us between the end of one dispatch and the start of
the next, with the kernels on both sides.    python tools/trace_gaps.py <kernel_trace.csv> [min_us=15]"""
import csv
import sys


def main():
    path = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ends = [i for i, r in enumerate(rows) if 'oks_nms' in r['Kernel_Name']]
    lo, hi = ends[-2], ends[-1]
    step = rows[lo:hi + 1]          # from the previous step's last kernel to this step's last kernel
    t0 = int(step[0]['End_Timestamp'])
    busy = 0
    gaps = []
    last_end = t0
    for prev, cur in zip(step[:-1], step[1:]):
        s, e = int(cur['Start_Timestamp']), int(cur['End_Timestamp'])
        gap = s - last_end
        if gap > 0:
            gaps.append((gap / 1e3, (last_end - t0) / 1e6, prev['Kernel_Name'][:70], cur['Kernel_Name'][:70]))
        busy += e - max(s, last_end) if e > last_end else 0
        last_end = max(last_end, e)
    wall = (last_end - t0) / 1e6
    print(f'# step wall {wall:.3f} ms, busy {busy / 1e6:.3f} ms, idle {wall - busy / 1e6:.3f} ms in {len(gaps)} gaps')
    big = [g for g in gaps if g[0] >= min_us]
    print(f'# gaps >= {min_us} us: {len(big)}, {sum(g[0] for g in big) / 1e3:.3f} ms; smaller: '
          f'{len(gaps) - len(big)}, {sum(g[0] for g in gaps if g[0] < min_us) / 1e3:.3f} ms')
    for g in big:
        print(f'{g[1]:8.3f} ms  gap {g[0]:8.1f} us   after {g[2]}\n{"":34s}before {g[3]}')


if __name__ == '__main__':
    main()
