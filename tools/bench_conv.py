"""Timing of the trunk's channel-major convolution kernel in isolation, next to fill_/copy_ of the same bytes."""
import sys

import torch

sys.path.insert(0, ".")
from geoa3_amd import _lib  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    lib = _lib.load()
    B, N = 250, 1024
    for K, Co, gate in [(64, 64, False), (64, 64, True), (64, 128, False), (128, 64, True)]:
        X = torch.randn(B, K, N, device="cuda")
        W = torch.randn(Co, K, device="cuda") * 0.1
        bias = torch.randn(Co, device="cuda")
        Z = torch.randn(B, Co, N, device="cuda") if gate else None
        Y = torch.empty(B, Co, N, device="cuda")
        ref = torch.relu(torch.einsum("ok,bkn->bon", W, X) + bias.view(1, -1, 1))
        if gate:
            ref = ref * (Z > 0)
        mb = (X.numel() + Y.numel() + (Z.numel() if gate else 0)) * 4 / 1e6
        s = torch.cuda.current_stream().cuda_stream
        print("   fill_ of Y: %.1f us; copy_ Y<-Y2: %.1f us" % (timeit(lambda: Y.fill_(1.0)),
                                                              timeit(lambda: Y.copy_(ref))))
        ref64 = torch.relu(torch.einsum("ok,bkn->bon", W.double(), X.double()) + bias.double().view(1, -1, 1))
        if gate:
            ref64 = ref64 * (Z > 0)
        for split in (0, 1):
            fn = lambda: lib.geoa3_debug_conv_cm(X.data_ptr(), W.data_ptr(), bias.data_ptr(),
                                                 Z.data_ptr() if gate else None, Y.data_ptr(), B, N, K, Co, 1, split, s)
            us = timeit(fn)
            err = (Y.double() - ref64).abs().max().item()
            print("K=%3d Co=%3d gate=%d %s: %6.1f us  %5.2f TB/s  (%.0f MB)  max err vs float64 %.2e"
                  % (K, Co, gate, "split fp16" if split else "fp32 MFMA ", us, mb / us, mb, err))


if __name__ == "__main__":
    main()
