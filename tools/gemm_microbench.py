#!/usr/bin/env python
"""Micro-benchmark of the grouped fp32-MFMA GEMM through set_linear_f32 (GPU box only).

    python tools/gemm_microbench.py [M N K ...]          # default: the decode-step shapes
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from show_edit_tell_amd.editnet import _HipLinear

SHAPES = [(128, 4096, 3072), (128, 4096, 1024), (128, 1024, 1024), (128, 10000, 1024), (128, 4096, 6144),
          (2560, 4096, 1024), (4608, 1024, 2048), (4608, 512, 1024), (4, 4096, 3072), (64, 4096, 3072)]


def main():
    shapes = SHAPES
    if len(sys.argv) > 3:
        a = list(map(int, sys.argv[1:]))
        shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
    dev = torch.device("cuda:0")
    iters = int(os.environ.get("ITERS", "50"))
    for M, N, K in shapes:
        lin = _HipLinear(K, N).to(dev)
        x = torch.rand(M, K, device=dev) * 2 - 1
        with torch.no_grad():
            for _ in range(5):
                y = lin(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                y = lin(x)
            e1.record()
            torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        fl = 2.0 * M * N * K
        ref = (x.double() @ lin.weight.double().t() + lin.bias.double())
        err = (y.double() - ref).abs().max().item()
        print("M=%5d N=%6d K=%5d  %8.2f us  %7.2f TFLOP/s  %7.1f GB/s(W)  maxerr %.1e" %
              (M, N, K, us, fl / us / 1e6, 4.0 * N * K / us / 1e3, err), flush=True)


if __name__ == "__main__":
    main()
