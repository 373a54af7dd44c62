"""Timing of the sparse arg-max backward kernel in isolation."""
import sys

import torch

sys.path.insert(0, ".")
from geoa3_amd import _lib  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402


def main():
    lib = _lib.load()
    N = 1024
    s = torch.cuda.current_stream().cuda_stream
    for B, taps in ((250, 1), (250, 3), (32, 1), (32, 3)):
        g = torch.randn(B, 1024, device="cuda")
        # arg-max columns concentrated on ~35 % of the points, like a trained max-pool
        pool = torch.randint(0, N, (B, 360), device="cuda")
        arg = torch.gather(pool, 1, torch.randint(0, 360, (B, 1024), device="cuda")).int().contiguous()
        W = torch.randn(1024, taps * 128, device="cuda") * 0.05
        Z = torch.randn(B, 128, N, device="cuda")
        dX = torch.empty(B, 128, N, device="cuda")
        res = {}
        for form in (1, 0):
            fn = lambda: lib.geoa3_debug_wide_bwd(g.data_ptr(), arg.data_ptr(), W.data_ptr(), Z.data_ptr(),
                                                  dX.data_ptr(), B, N, taps, form, s)
            us = timeit(fn)
            res[form] = dX.clone()
            print("B=%d taps=%d form=%d: %6.1f us  %5.2f TB/s (write + gate read)" % (B, taps, form, us, 2 * B * 128 * N * 4e-6 / us))
        print("   forms bit-identical:", bool(torch.equal(res[0], res[1])))


if __name__ == "__main__":
    main()
