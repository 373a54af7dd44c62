"""Timing of one fully connected layer (fc_kernel) in isolation: python tools/bench_fc.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from geoa3_amd import _lib  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402


def main():
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    M = 250
    for K, Nout in [(1024, 512), (512, 256), (256, 9), (256, 4096), (256, 40), (40, 256), (256, 512), (512, 1024),
                    (4096, 256), (128, 512), (2048, 512)]:
        X = torch.randn(M, K, device="cuda")
        W = torch.randn(Nout, K, device="cuda") * 0.05
        bias = torch.randn(Nout, device="cuda")
        Y = torch.empty(M, Nout, device="cuda")
        ref = torch.relu(X @ W.t() + bias)
        for ks in (0, 8):
            us = timeit(lambda: lib.geoa3_debug_fc(X.data_ptr(), W.data_ptr(), bias.data_ptr(), Y.data_ptr(), M, Nout, K,
                                                   1, ks, s), iters=50)
            err = float((Y - ref).abs().max())
            print("K=%4d Nout=%4d ksplit=%d: %6.1f us  (workgroups %d)  maxerr %.1e"
                  % (K, Nout, ks, us, ((Nout + 31) // 32) * 8, err))


if __name__ == "__main__":
    main()
