"""Timing of one fully connected layer (fc_kernel) in isolation: python tools/bench_fc.py
variant = (tile << 8) | waves splitting K; 0 = the shipped choice (launch_fc)."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from geoa3_amd import _lib  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402


def main():
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    for M in (32, 250):
        for K, Nout in [(1024, 512), (512, 256), (256, 9), (256, 4096), (256, 40), (40, 256), (9, 256), (256, 512),
                        (512, 1024), (4096, 256)]:
            X = torch.randn(M, K, device="cuda")
            W = torch.randn(Nout, K, device="cuda") * 0.05
            bias = torch.randn(Nout, device="cuda")
            Y = torch.empty(M, Nout, device="cuda")
            ref = torch.relu(X.double() @ W.double().t() + bias.double())
            line = "M=%3d K=%4d Nout=%4d:" % (M, K, Nout)
            for tile, waves in ((0, 0), (16, 4), (16, 8), (16, 16), (32, 4), (32, 8), (32, 16)):
                v = (tile << 8) | waves
                us = timeit(lambda: lib.geoa3_debug_fc(X.data_ptr(), W.data_ptr(), bias.data_ptr(), Y.data_ptr(), M,
                                                       Nout, K, 1, v, s), iters=50)
                err = float((Y.double() - ref).abs().max())
                line += "  t%02dw%02d %5.1f us (%.0e)" % (tile, waves, us, err)
            print(line)


if __name__ == "__main__":
    main()
