#!/usr/bin/env python
"""gemm_gen_asm (hand-written k-loop of the 128x64 general-layout GEMM) against the compiler-scheduled kernel: the results must
be BIT-IDENTICAL (same tile, same K permutation, same MFMA order) for both layouts, ragged edges and split-K included.
Needs a library built with SET_HIPCC_FLAGS=-DSET_EXPERIMENTAL_GEMMS (python -m show_edit_tell_amd.build --force)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def run():
    from show_edit_tell_amd import autograd_ops as A
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    out = {}
    # (name, a_kminor, M, N, K): dW = dY^T X (a k-minor) and dX = dY W (a k-major), B k-minor in both
    for name, akm, M, N, K in [("dW x2h", True, 4096, 4096, 2432), ("dW ragged", True, 9490, 1024, 2432), ("dW small", True, 1024, 512, 2432),
                                ("dW Ktail", True, 1024, 1024, 1000), ("dX fc", False, 2432, 1024, 9472), ("dX att", False, 87552 // 8, 1024, 512),
                                ("dX rag", False, 2400, 1020, 4096), ("dX one", False, 640, 64, 32)]:
        if akm:
            a = torch.randn(K, M + (4 - M % 4) % 4, device=dev)[:, :M] if M % 4 else torch.randn(K, M, device=dev)
            b = torch.randn(K, N, device=dev)
            c = A.gemm(a, True, b, True, M, N, K)
        else:
            a = torch.randn(M, K, device=dev)
            b = torch.randn(K, N, device=dev)
            c = A.gemm(a, False, b, True, M, N, K)
        out[name] = c.cpu()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1:
        torch.save(run(), sys.argv[1])
        sys.exit(0)
    res = []
    for v in ("1", "0"):
        f = "/tmp/gga_%s.pt" % v
        subprocess.run([sys.executable, __file__, f], env=dict(os.environ, SET_GEMM_GEN_ASM=v), check=True)
        res.append(torch.load(f))
    ok = True
    for k in res[0]:
        same = torch.equal(res[0][k], res[1][k])
        ok = ok and same
        print("%-10s asm == compiler-scheduled: %s   (max |diff| %.3e)" % (k, same, float((res[0][k] - res[1][k]).abs().max())))
    print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
