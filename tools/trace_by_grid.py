"""Average kernel duration per (kernel, grid) from a rocprofv3 kernel trace CSV:  python tools/trace_by_grid.py trace.csv [name filter]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(list)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if flt and flt not in name:
        continue
    key = (name[:70], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Workgroup_Size_X"))
    acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items()):
    v.sort()
    print(f"{k[0]:70s} grid {k[1]:>7s}x{k[2]:>5s} wg {k[3]:>4s}  n {len(v):4d}  avg {sum(v)/len(v)/1e3:7.2f} us  med {v[len(v)//2]/1e3:7.2f}  min {v[0]/1e3:7.2f}")
