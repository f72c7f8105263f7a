"""Median duration per (kernel, grid, workgroup) of a rocprofv3 kernel trace, launch order kept:
python3 tools/trace_groups.py <kernel_trace.csv>"""
import csv
import re
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
groups, order = {}, []
for r in rows:
    name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))
    key = (name[:48], r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?')))
    if key not in groups:
        groups[key] = []
        order.append(key)
    groups[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in order:
    d = groups[k]
    if len(d) >= 10:
        print(f'{k[0]:50s} grid {k[1]:>8s} wg {k[2]:>5s}  n={len(d):4d}  median {statistics.median(d):7.1f} us  min {min(d):7.1f}')
