"""Idle gaps of the main queue in a rocprofv3 --kernel-trace of bench.py: gap_analysis.py <dir with *kernel_trace.csv>."""
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
mainq = [r['Queue_Id'] for r in rows if 'winograd_kernel' in r['Kernel_Name']][0]
main = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows if r['Queue_Id'] == mainq)
w = [i for i, m in enumerate(main) if 'winograd_kernel' in m[2]]
seg = main[w[len(w) // 4]:w[-1] + 1]          # skip the warm-up point
span = (seg[-1][1] - seg[0][0]) / 1e6
busy = sum(e - s for s, e, _ in seg) / 1e6
gaps = [(seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)]
print(f"main queue {mainq}: {len(seg)} kernels over {span:.1f} ms, busy {busy:.1f} ms, idle {span - busy:.1f} ms")
for g, a, b in sorted(((g, seg[i][2][:48], seg[i + 1][2][:48]) for i, g in enumerate(gaps)), reverse=True)[:8]:
    print(f"  gap {g:9.1f} us after {a} before {b}")
side = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows if r['Queue_Id'] != mainq)
if side:
    t0, t1 = seg[0][0], seg[-1][1]
    ss = [x for x in side if t0 <= x[0] <= t1]
    print(f"side queue(s): {len(ss)} kernels in that span, busy {sum(e - s for s, e, _ in ss) / 1e6:.1f} ms")
