"""Per training step (from one `zero_ranges_kernel` to the next) in a rocprofv3 kernel trace: wall time, time with at least one
kernel running, idle time, and the kernels the largest idle gaps precede.  usage: step_gaps.py <dir> [first_kernel_prefix]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
first = sys.argv[2] if len(sys.argv) > 2 else "zero_ranges_kernel"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(first)]
for a, b in list(zip(starts, starts[1:]))[-3:]:
    seg = rows[a:b]
    t0 = int(seg[0]["Start_Timestamp"])
    wall = int(rows[b]["Start_Timestamp"]) - t0
    busy, end, gaps = 0, t0, []
    small = collections.Counter()
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > end:
            gaps.append((s - end, r["Kernel_Name"][:60]))
            busy += e - s
        elif e > end:
            busy += e - end
        end = max(end, e)
    idle = wall - busy
    gaps.sort(reverse=True)
    by = collections.Counter()
    for g, n in gaps:
        by[n] += g
    print(f"step: {len(seg)} kernels, wall {wall / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {idle / 1e3:.1f} us in {len(gaps)} gaps "
          f"(median {sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3:.2f} us)")
    print("   largest:", [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:6]])
    print("   idle by following kernel:", [(round(g / 1e3, 1), n[:40]) for n, g in by.most_common(8)])
