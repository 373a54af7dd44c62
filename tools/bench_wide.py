"""Timing of the 1024-wide forward layers in isolation, fp32 MFMA against the split-fp16 kernel (and its tuning
variants): python tools/bench_wide.py [variants...]"""
import sys

import torch

import os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from geoa3_amd import _lib  # noqa: E402
from geoa3_amd.pointnet import pack_wide_fragments, pack_wide_split  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402


def main():
    lib = _lib.load()
    variants = [int(v) for v in sys.argv[1:] if v != "--stamps"] or [0]
    want_stamps = "--stamps" in sys.argv
    B, N = 250, 1024
    s = torch.cuda.current_stream().cuda_stream
    for taps in (3, 1):
        X = torch.randn(B, 128, N, device="cuda").relu_()
        W = torch.randn(1024, taps * 128) * 0.05
        Wp = pack_wide_fragments(W, taps).cuda()
        Wh, uns = pack_wide_split(W, taps)
        Wh = Wh.cuda()
        bias = torch.randn(1024, device="cuda")
        out = torch.empty(B, 1024, device="cuda")
        arg = torch.empty(B, 1024, device="cuda", dtype=torch.int32)
        keys = torch.empty(B, 1024, device="cuda", dtype=torch.int64)
        flops = 2.0 * B * N * 1024 * 128 * taps

        def run(split, variant, stamps=None):
            return lib.geoa3_debug_wide_fwd(X.data_ptr(), Wp.data_ptr(), Wh.data_ptr() if split else None, uns,
                                            bias.data_ptr(), out.data_ptr(), arg.data_ptr(), keys.data_ptr(), B, N,
                                            taps, variant, stamps, s)
        us = timeit(lambda: run(False, 0))
        ref = out.clone()
        print("taps=%d fp32 MFMA      : %7.1f us  %6.1f TF" % (taps, us, flops / us / 1e6))
        for v in variants:
            us = timeit(lambda: run(True, v))
            err = float((out - ref).abs().max() / ref.abs().max())
            if want_stamps:
                st = torch.zeros(256, device="cuda", dtype=torch.int64)
                run(True, v, st.data_ptr())
                torch.cuda.synchronize()
                st = st.cpu().tolist()
                t = st[1:1 + st[0]]
                print("   s_memtime deltas of workgroup 0 / wave 0 (cycles):", [t[i + 1] - t[i] for i in range(len(t) - 1)])
            print("taps=%d split variant %d: %7.1f us  %6.1f TF-equivalent (%.0f TF on the f16 pipe)  max rel diff %.1e"
                  % (taps, v, us, flops / us / 1e6, 3 * flops / us / 1e6, err))


if __name__ == "__main__":
    main()
