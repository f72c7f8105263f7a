"""Kernel sequence (launch order, duration) of the LAST steady-state step of a rocprofv3
--kernel-trace CSV; steps are delimited by the oks_nms kernel.  Used to attribute library kernel
names to model stages.    python tools/trace_sequence.py <kernel_trace.csv> [min_us=50]"""
import csv
import re
import sys


def short(n):
    m = re.search(r'MT\d+x\d+x\d+', n)
    if n.startswith('Cijk'):
        return 'hipblaslt ' + m.group(0)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ends = [i for i, r in enumerate(rows) if 'oks_nms' in r['Kernel_Name']]
    sel = rows[ends[-2] + 1:ends[-1] + 1]
    t0 = int(sel[0]['Start_Timestamp'])
    for r in sel:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if d >= min_us:
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.2f} ms  {d:8.1f} us  "
                  f"grid {r.get('Grid_Size', '?'):>10s}  {short(r['Kernel_Name'])}")


if __name__ == '__main__':
    main()
