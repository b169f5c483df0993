"""Times the training step's large contractions one by one (same entry points as the sequence node):
   wgrad dW = dY^T X (time-batched), the va_fa forward / dgrad over all regions, fc forward over all timesteps."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from show_edit_tell_amd import autograd_ops as A

dev = torch.device("cuda:0")
TB, R = 19 * 128, 36


def tm(fn, n=10, w=3):
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


rows = []
for name, N, K, M in [("fc_w", 9490, 1024, TB), ("x2h", 4096, 4096, TB), ("h2h", 4096, 1024, TB), ("al_blk", 4096, 1024, TB),
                      ("gate", 1024, 3072, TB), ("tc", 1024, 2048, TB), ("sc", 1024, 1024, TB), ("dec", 512, 1024, TB),
                      ("va_fa", 512, 1024, TB * R), ("va_fa_step", 512, 1024, 128 * R)]:
    dy = torch.randn(M, N, device=dev)
    x = torch.randn(M, K, device=dev)
    out = torch.zeros(N, K, device=dev)
    us = tm(lambda: A._wgrad_mm(dy, x, out=out))
    rows.append(("wgrad " + name, N, K, M, us, 2.0 * N * K * M / us / 1e6))
# forward / dgrad of features_att over one step's regions (128*36 rows)
x = torch.randn(128 * R, 1024, device=dev); w = torch.randn(512, 1024, device=dev); b = torch.zeros(512, device=dev)
y = torch.empty(128 * R, 512, device=dev)
from show_edit_tell_amd.xe_sequence import _Ops
ops = _Ops(dev)
us = tm(lambda: ops.linear(x, w, b, y, 128 * R)); rows.append(("fwd va_fa step", 512, 1024, 128 * R, us, 2.0 * 512 * 1024 * 128 * R / us / 1e6))
dx = torch.empty(128 * R, 1024, device=dev)
us = tm(lambda: A.gemm(y, False, w, True, 128 * R, 1024, 512, out=dx)); rows.append(("dgrad va_fa step", 1024, 512, 128 * R, us, 2.0 * 512 * 1024 * 128 * R / us / 1e6))
h = torch.randn(TB, 1024, device=dev); wf = torch.randn(9490, 1024, device=dev); bf = torch.zeros(9490, device=dev); p = torch.empty(TB, 9490, device=dev)
us = tm(lambda: ops.linear(h, wf, bf, p, TB)); rows.append(("fwd fc all t", 9490, 1024, TB, us, 2.0 * 9490 * 1024 * TB / us / 1e6))
dh = torch.empty(TB, 1024, device=dev)
us = tm(lambda: A._dgrad(p, wf)); rows.append(("dgrad fc all t", 1024, 9490, TB, us, 2.0 * 9490 * 1024 * TB / us / 1e6))
for r in rows:
    print("%-18s N=%5d K=%5d M=%6d  %8.1f us  %6.1f TFLOP/s" % r)
