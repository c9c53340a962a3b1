"""Where one steady-state bench point goes: point_budget.py <dir with *kernel_trace.csv>.
Takes the main queue between the first Winograd launch of the second-to-last point and the last launch of the trace,
and prints kernel time by kernel name (main queue and side queue), the span, and the idle time."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
mainq = [r['Queue_Id'] for r in rows if 'winograd_kernel' in r['Kernel_Name']][0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
main = [e for e in ev if e[3] == mainq]
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 4          # points in the trace (warm-up + timed)
w = [i for i, m in enumerate(main) if 'winograd_kernel' in m[2]]
per = len(w) // npts
a, b = main[w[per * (npts - 2)]][0], main[w[per * (npts - 1)]][0]      # one whole point: first conv of point n-2 .. first conv of point n-1
span = (b - a) / 1e6
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:60]
for q, label in ((True, "main queue"), (False, "other queues")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, qq in ev:
        if (qq == mainq) == q and a <= s < b:
            acc[short(n)][0] += 1; acc[short(n)][1] += (e - s) / 1e6
    tot = sum(v[1] for v in acc.values())
    print(f"{label}: {sum(v[0] for v in acc.values())} launches, {tot:.1f} ms of kernel time in a span of {span:.1f} ms" + (f" (idle {span - tot:.1f} ms)" if q else ""))
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"   {t:8.2f} ms  x{c:<5d} {n}")
