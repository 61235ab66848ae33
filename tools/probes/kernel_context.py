"""Kernels around the occurrences of one kernel in a rocprofv3 kernel trace: kernel_context.py <dir> <substring> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat, n = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
hits = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
for i in hits[-2:]:
    print("----")
    for j in range(max(0, i - n), min(len(rows), i + n + 1)):
        r = rows[j]
        print(("=> " if j == i else "   ") + f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  stream {r.get('Stream_Id')}  {r['Kernel_Name'][:90]}")
