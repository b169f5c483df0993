"""Sweeps the split-K depth of the grouped dX = dY.W launches of one training timestep (experiment knob SET_EXP_GEN_KPER)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from show_edit_tell_amd import autograd_ops as A

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "128"))


def tm(fn, n=30, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


groups = {"du(K1024: 2xN1024)": (B, 1024, [1024, 1024]), "dgw(K4096: N1024,1024,2048,1024)": (B, 4096, [1024, 1024, 2048, 1024]),
          "dszt(K2048: 3xN1024)": (B, 2048, [1024] * 3), "datt2(K1024: N1024)": (B, 1024, [1024]),
          "dg1(K4096: 3xN1024)": (B, 4096, [1024] * 3), "va_fa dgrad (M=B*36, K512, N1024)": (B * 36, 512, [1024]),
          "fc dgrad all t (M=19B, K9492, N1024)": (19 * B, 9492, [1024])}
for name, (M, K, Ns) in groups.items():
    dy = torch.randn(M, K, device=dev)
    ws = [torch.randn(K, N, device=dev) for N in Ns]
    outs = [torch.zeros(M, N, device=dev) for N in Ns]
    items = [(dy, w, M, w.shape[1], K, o, False) for w, o in zip(ws, outs)]
    res = []
    for kper in [0, 4, 6, 8, 11, 13, 16, 19, 22, 26, 32, 43, 64, 75, 100, 128, 150, 999]:
        if kper:
            os.environ["SET_EXP_GEN_KPER"] = str(kper)
        else:
            os.environ.pop("SET_EXP_GEN_KPER", None)
        if kper and kper < 999 and kper > K // 32:
            continue
        res.append((kper, tm(lambda: A.gemm_group(items, False, True))))
    os.environ.pop("SET_EXP_GEN_KPER", None)
    print(name, "  ".join("%s:%.1f" % (("auto" if k == 0 else ("none" if k == 999 else k)), t) for k, t in res))
