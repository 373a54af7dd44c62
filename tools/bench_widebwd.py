"""Timing of the sparse arg-max backward kernel in isolation."""
import sys

import torch

sys.path.insert(0, ".")
from geoa3_amd import _lib  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402


def main():
    lib = _lib.load()
    B, N = 250, 1024
    s = torch.cuda.current_stream().cuda_stream
    for taps in (1, 3):
        g = torch.randn(B, 1024, device="cuda")
        # arg-max columns concentrated on ~35 % of the points, like a trained max-pool
        pool = torch.randint(0, N, (B, 360), device="cuda")
        arg = torch.gather(pool, 1, torch.randint(0, 360, (B, 1024), device="cuda")).int().contiguous()
        W = torch.randn(1024, taps * 128, device="cuda") * 0.05
        Z = torch.randn(B, 128, N, device="cuda")
        dX = torch.empty(B, 128, N, device="cuda")
        fn = lambda: lib.geoa3_debug_wide_bwd(g.data_ptr(), arg.data_ptr(), W.data_ptr(), Z.data_ptr(),
                                              dX.data_ptr(), B, N, taps, s)
        us = timeit(fn)
        print("taps=%d: %6.1f us  %5.2f TB/s (262 MB)" % (taps, us, 262.1 / us))


if __name__ == "__main__":
    main()
