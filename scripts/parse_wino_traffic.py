"""parse_wino_traffic.py <log of wino_shapes.py> <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [43 | 43h | 1d]
(43: the F(4x4,3x3) kernel, winograd43_kernel / winograd43.hip, 36 transformed positions; 43h: its fp16-pair form, winograd43h_kernel /
winograd43h.hip; 1d: the row-wise F(4,3) pair kernel, wino1d_kernel / wino1d.hip, 18 transformed slots; default the F(2x2,3x3) kernel)
HBM bytes per launch = FETCH_SIZE * 1024 * 2 (gfx950 counts half of a wide coalesced read, MI355X_MICROARCH.md section
HBM) + WRITE_SIZE * 1024; the second isolated launch of each shape is taken."""
import csv, glob, hashlib, json, os, sys
log, fdir, wdir, out = sys.argv[1:5]
MODE = sys.argv[5] if len(sys.argv) > 5 else ""
KERNEL, NPOS, SRC = {"43": ("winograd43_kernel", 36, ["winograd43.hip", "winograd43_shared.h"]),
                     "43h": ("winograd43h_kernel", 36, ["winograd43h.hip", "winograd43_shared.h"]),
                     "1d": ("wino1d_kernel", 18, ["wino1d.hip"])}.get(MODE, ("winograd_kernel", 16, ["winograd.hip"]))
keys = [(l.split()[1], int(l.split()[2])) for l in open(log) if l.startswith("KEY")]

def counters(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]

fetch, write = counters(fdir, "FETCH_SIZE"), counters(wdir, "WRITE_SIZE")
n = len(keys)
fetch, write = fetch[-2 * n:], write[-2 * n:]
assert len(fetch) == 2 * n and len(write) == 2 * n, (len(fetch), len(write), n)
shapes = {}
for i, (k, cnt) in enumerate(keys):
    B, H, W, rest = k.split("x")
    Cin, Cout = rest.split("->")
    B, H, W, Cin, Cout = map(int, (B, H, W, Cin, Cout))
    fb, wb = fetch[2 * i + 1] * 1024 * 2, write[2 * i + 1] * 1024
    alg = 4 * (B * H * W * Cin + NPOS * Cin * Cout + B * H * W * Cout)
    shapes[k] = {"fetch_bytes": fb, "write_bytes": wb, "total_bytes": fb + wb, "algorithmic_bytes": alg,
                 "ratio": round((fb + wb) / alg, 3), "calls_per_forward": cnt}
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "id-diff_amd", "csrc")
sha = hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in SRC)).hexdigest()      # bench.py reports traffic only for these very sources
json.dump({"kernel": KERNEL, "kernel_source_sha256": sha, "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on scripts/wino_shapes.py: "
                   "every distinct Winograd-conv call of the nf=128 NCSN++ forward at the bench's launch-set size, second "
                   "isolated launch of each; bytes = FETCH_SIZE*1024*2 (gfx950 correction) + WRITE_SIZE*1024; algorithmic = "
                   "input + transformed filters + output, each once", "shapes": shapes}, open(out, "w"), indent=1)
for k, v in shapes.items():
    print(k, v["ratio"], f'{v["total_bytes"]/1e9:.3f} GB')
