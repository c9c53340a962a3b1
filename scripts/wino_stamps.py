"""In-kernel stamps of the Winograd kernel (diagnostic launch, IDIFF_WINO_DBG=4): per-workgroup phase durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import id_diff_amd
from id_diff_amd import _lib
dev = torch.device("cuda:0")
B = 2240
for H, Cin, Cout in [(32, 128, 128), (16, 512, 256)]:
    x = torch.randn(B, H * H, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
    u = _lib.winograd_pack(w, Cin, Cout)
    o = torch.empty(B, H * H, Cout, device=dev)
    nwg = (B * H * H // 4 // 64) * (Cout // 64)
    st = torch.zeros(nwg * 6 + 4096 * 32, device=dev, dtype=torch.float64)
    for _ in range(3):
        _lib.conv2d_winograd(x, u, o, B, H, H, Cin, Cout, epilogue=_lib.make_epilogue(bias=torch.zeros(Cout, device=dev)))
    torch.cuda.synchronize()
    os.environ["IDIFF_WINO_DBG"] = "4"
    _lib.conv2d_winograd(x, u, o, B, H, H, Cin, Cout, epilogue=_lib.make_epilogue(bias=torch.zeros(Cout, device=dev), rows_per_group=H * H, colstats=st))
    torch.cuda.synchronize()
    os.environ["IDIFF_WINO_DBG"] = "0"
    raw = st.cpu().numpy().view(np.int64)
    a = raw[:nwg * 6].reshape(nwg, 6)
    ph = raw[nwg * 6:].reshape(4096, 8, 4)[: min(4096, nwg)]
    steps = Cin // 8
    for wv in range(8):
        m = np.median(ph[:, wv, :], axis=0) / steps
        print(f"   wave {wv} ({'V' if wv < 4 else 'U'}): per step stage {m[0]:.0f} fetch {m[1]:.0f} mfma {m[2]:.0f} barrier {m[3]:.0f}  sum {m.sum():.0f}", flush=True)
    pro, loop, tail = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
    tot = a[:, 3] - a[:, 0]
    print(f"{H}x{H} {Cin}->{Cout}: {nwg} WGs; cycles median: prologue {np.median(pro):.0f}  loop {np.median(loop):.0f} ({np.median(loop)/(Cin//8):.0f}/step)  "
          f"tail {np.median(tail):.0f}  total {np.median(tot):.0f}", flush=True)
    # gap between consecutive workgroups on one CU: sort by start within the same hw id
    hw = a[:, 5]
    gaps = []
    for cu in np.unique(hw)[:64]:
        sel = a[hw == cu]
        sel = sel[np.argsort(sel[:, 0])]
        gaps += list(sel[1:, 0] - sel[:-1, 3])
    gaps = np.array(gaps)
    print(f"   launch gap (end of a WG -> start of the next with the same HW_ID) median {np.median(gaps):.0f} cycles, "
          f"p10 {np.percentile(gaps,10):.0f} p90 {np.percentile(gaps,90):.0f}; wall {(a[:,4].max()-a[:,4].min())/100:.0f} us", flush=True)
