import sys, torch
sys.path.insert(0, ".")
from geoa3_amd import _lib
from tools.bench_conv import timeit
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
for B in (2, 32):
    N = 1024
    for K, Co in ((64, 64), (64, 128), (128, 64)):
        X = torch.randn(B, K, N, device="cuda"); W = torch.randn(Co, K, device="cuda") * 0.1
        bias = torch.randn(Co, device="cuda"); Z = torch.randn(B, Co, N, device="cuda"); Y = torch.empty(B, Co, N, device="cuda")
        for name, b_, z_, r_ in (("plain", None, None, 0), ("bias+relu", bias, None, 1), ("gate", None, Z, 0), ("bias+relu+gate", bias, Z, 1)):
            fn = lambda: lib.geoa3_debug_conv_cm(X.data_ptr(), W.data_ptr(), b_.data_ptr() if b_ is not None else None,
                                                 z_.data_ptr() if z_ is not None else None, Y.data_ptr(), B, N, K, Co, r_, 1, s)
            print("B=%d K=%d Co=%d %-15s %.1f us" % (B, K, Co, name, timeit(fn, iters=50)))
