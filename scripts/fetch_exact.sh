#!/bin/bash
# Exact fabric read bytes by request size (TCC_EA0_RDREQ_{32B,64B,128B}): first on the calibration kernels of scripts/fetch_calib.hip (known
# byte counts), then on every distinct pair-convolution call of the forward (scripts/wino_shapes.py 2240 43h)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_fetch_exact; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SET="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum"
timeout -k 10 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/calib -- $R/scripts/fetch_calib > $O/calib.log 2>&1 || echo "calib failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/shapes -- python3 $R/scripts/wino_shapes.py 2240 43h > $O/shapes.log 2>&1 || echo "shapes failed"
python3 - <<PY | tee $R/gpurun_out/r05_fetch_exact.txt
import csv, glob, collections, json
def load(d):
    acc = collections.defaultdict(dict)
    names = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    return acc, names
acc, names = load("$O/calib")
print("calibration (bytes = 32 n32 + 64 n64 + 128 n128; n = TCC_EA0_RDREQ_*):")
asked = [256.0, 256.0, 128.0, 32.0]
ids = sorted(acc)[-4:]
for d, a in zip(ids, asked):
    c = acc[d]
    n32, n64, n128, n = c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0), c.get("TCC_EA0_RDREQ_128B_sum", 0), c.get("TCC_EA0_RDREQ_sum", 0)
    print(f"  {names[d].split('(')[0]:10s} asked {a:6.1f} MiB: n32 {n32:.3g} n64 {n64:.3g} n128 {n128:.3g} all {n:.3g} -> {(32*n32+64*n64+128*n128)/2**20:7.1f} MiB by sizes; other-size requests {n-n32-n64-n128:.3g}")
acc, names = load("$O/shapes")
keys = [l.split()[1] for l in open("$O/shapes.log") if l.startswith("KEY")]
conv = [d for d in sorted(acc) if "winograd43h_kernel" in names[d]]
iso = conv[-2 * len(keys):]
out = {}
print("winograd43h_kernel, second isolated launch of each shape: exact fabric read bytes")
for i, k in enumerate(keys):
    c = acc[iso[2 * i + 1]]
    n32, n64, n128, n = c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0), c.get("TCC_EA0_RDREQ_128B_sum", 0), c.get("TCC_EA0_RDREQ_sum", 0)
    b = 32*n32 + 64*n64 + 128*n128
    out[k] = {"n32": n32, "n64": n64, "n128": n128, "n": n, "fetch_bytes_exact": b}
    print(f"  {k:24s} {b/1e6:9.1f} MB  (n64 {n64:.3g}, n128 {n128:.3g}, n32 {n32:.3g}; FETCH_SIZE would report {n*64/1e6:9.1f} MB, x2 = {n*128/1e6:9.1f})")
json.dump(out, open("$R/gpurun_out/r05_fetch_exact.json", "w"), indent=1)
PY
