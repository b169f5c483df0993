"""Persistent small-batch EditNet decode (csrc/decode_persistent_wide.hip) against the per-step loop.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import numpy as np, torch
import parity
from hip_adapter import editnet_modules, to_dev
from show_edit_tell_amd import _lib

d, xe, rl = editnet_modules("editnet_full_b4")
g = parity.load("editnet_full_b4")
prev, plen, X = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"])
lib = _lib.load()


def run(p, l, x, n=20):
    with torch.no_grad():
        for _ in range(3):
            out = rl(d["wm"], p, l, x, True, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = rl(d["wm"], p, l, x, True, False)
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0) / n * 1e3


for B in [int(x) for x in os.environ.get("PROBE_BS", "4,1,2,3,8").split(",")]:
    if B == 4:
        p, l, x = prev, plen, X
    else:
        rs = np.random.RandomState(B)
        T = prev.shape[1]
        l_np = rs.randint(1, T + 1, size=(B, 1)).astype(np.int64)
        p_np = rs.randint(4, 9000, size=(B, T)).astype(np.int64)
        for i in range(B):
            p_np[i, l_np[i, 0]:] = 0
        p, l = to_dev(p_np), to_dev(l_np)
        x = to_dev(np.abs(rs.randn(B, X.shape[1], X.shape[2])).astype(np.float32))
    os.environ["SET_DEC_PERSISTENT"] = "0"
    ref, ms0 = run(p, l, x)
    os.environ["SET_DEC_PERSISTENT"] = "1"
    lib.set_profile_enable(1)
    with torch.no_grad():
        rl(d["wm"], p, l, x, True, False)
    torch.cuda.synchronize()
    tags = [r["tag"] for r in _lib.profile_report()]
    lib.set_profile_enable(0)
    out, ms1 = run(p, l, x)
    same = (out[0] == ref[0]).all(1).cpu().numpy()
    dl = float((out[1] - ref[1]).abs()[torch.from_numpy(same).to(out[1].device)].max()) if same.any() else float("nan")
    print("B=%2d per-step %.3f ms  persistent %.3f ms  used=%s  rows equal %d/%d  max|dlogp| %.2e" %
          (B, ms0, ms1, "persistent_decode" in tags, int(same.sum()), B, dl), flush=True)
    if not same.all():
        print(out[0].cpu().numpy()[:, :8], ref[0].cpu().numpy()[:, :8], out[1].cpu().numpy()[:, :4], ref[1].cpu().numpy()[:, :4])
    if B == 4:
        parity.check_greedy(out[0].cpu().numpy(), out[1].cpu().numpy(), g)
        print("golden editnet_full_b4 OK (persistent)")
