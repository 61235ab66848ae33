"""Runs of `__amd_rocclr_copyBuffer` in a rocprofv3 kernel trace: length, what ran before / after.  usage: copy_neighbours.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"][:70] for r in rows]
i = 0
while i < len(names):
    if "copyBuffer" in names[i]:
        j = i
        while j < len(names) and "copyBuffer" in names[j]:
            j += 1
        t0, t1 = int(rows[i]["Start_Timestamp"]), int(rows[j - 1]["End_Timestamp"])
        gap_b = t0 - int(rows[i - 1]["End_Timestamp"]) if i else 0
        print(f"run of {j - i:3d} copies, {1e-3 * (t1 - t0):8.1f} us; before: {names[i - 1] if i else '-'} (gap {1e-3 * gap_b:.1f} us); "
              f"after: {names[j] if j < len(names) else '-'}")
        i = j
    else:
        i += 1
