"""TEST HELPER (not product): evaluates the PACKED PointNet weights (geoa3_amd.pointnet.pack_pointnet) with
plain torch CPU ops, following the same decomposition as the HIP launch sequence in
geoa3_amd/csrc/pointnet.hip -- folded BN, max fused after the wide layers, and the hand-derived sparse
input-gradient.  Lets the host-side packing and the backward derivation be checked without a GPU."""
import torch


def _wide(X, W, bias, taps):
    # X [B,128,N]; W [Co, taps*128] (k = tap*128+ci); returns relu(max_n z + bias), argmax n
    B, Ci, N = X.shape
    if taps == 1:
        z = torch.einsum("ok,bkn->bon", W, X)
    else:
        Xp = torch.nn.functional.pad(X, (1, 1))
        z = sum(torch.einsum("ok,bkn->bon", W[:, t * Ci:(t + 1) * Ci], Xp[:, :, t:t + N]) for t in range(3))
    m, arg = z.max(-1)
    return torch.relu(m + bias), arg


def _wide_bwd(g, arg, W, Z, taps):
    # dX[b][ci][m] = sum_{co,tap: arg+tap-taps//2==m} W[co][tap*128+ci] g[b][co], gated by Z>0
    B, Co = g.shape
    Ci, N = Z.shape[1], Z.shape[2]
    dX = torch.zeros(B, Ci, N, dtype=g.dtype)
    for t in range(taps):
        m = arg + t - taps // 2
        ok = (m >= 0) & (m < N)
        contrib = g.unsqueeze(-1) * W[:, t * Ci:(t + 1) * Ci].unsqueeze(0)      # [B,Co,Ci]
        contrib = contrib * ok.unsqueeze(-1)
        idx = m.clamp(0, N - 1).unsqueeze(-1).expand(B, Co, Ci)
        dX.scatter_add_(2, idx.permute(0, 2, 1), contrib.permute(0, 2, 1))
    return dX * (Z > 0)


def _conv(X, W, b=None, relu=False):
    y = torch.einsum("ok,bkn->bon", W, X)
    if b is not None:
        y = y + b.view(1, -1, 1)
    return torch.relu(y) if relu else y


def _fc(X, W, b=None, relu=False):
    y = X @ W.t()
    if b is not None:
        y = y + b
    return torch.relu(y) if relu else y


def _tnet_fwd(t, x_or_feat, first):
    s = {}
    s["a1"] = first
    s["a2"] = _conv(s["a1"], t["w2"], t["b2"], True)
    s["p"], s["arg"] = _wide(s["a2"], t["w3"], t["b3"], 1)
    s["f4"] = _fc(s["p"], t["f1"], t["fb1"], True)
    s["f5"] = _fc(s["f4"], t["f2"], t["fb2"], True)
    s["T"] = _fc(s["f5"], t["f3"], t["fb3"])
    return s


def _tnet_bwd(t, s, gT):
    g = _fc(gT, t["f3t"]) * (s["f5"] > 0)
    g = _fc(g, t["f2t"]) * (s["f4"] > 0)
    g = _fc(g, t["f1t"]) * (s["p"] > 0)
    G128 = _wide_bwd(g, s["arg"], t["w3"], s["a2"], 1)
    return _conv(G128, t["w2t"]) * (s["a1"] > 0)      # d/d pre-activation of the first layer


def forward(p, x):
    """-> logits, saved"""
    B, _, N = x.shape
    S = {}
    S["t3"] = _tnet_fwd(p["t3"], x, _conv(x, p["t3"]["w1"], p["t3"]["b1"], True))
    T3 = S["t3"]["T"].view(B, 3, 3)
    xp = torch.einsum("bdc,bdn->bcn", T3, x)
    S["h1"] = _conv(xp, p["w1"], p["b1"], True)
    S["h2"] = _conv(S["h1"], p["w2"], p["b2"], True)
    S["t64"] = _tnet_fwd(p["t64"], S["h2"], _conv(S["h2"], p["t64"]["w1"], p["t64"]["b1"], True))
    T64 = S["t64"]["T"].view(B, 64, 64)
    h2p = torch.einsum("bij,bin->bjn", T64, S["h2"])
    S["h3"] = _conv(h2p, p["w3"], p["b3"], True)
    S["h4"] = _conv(S["h3"], p["w4"], p["b4"], True)
    S["p5"], S["i5"] = _wide(S["h4"], p["w5"], p["b5"], 3)
    S["f6"] = _fc(S["p5"], p["f1"], p["fb1"], True)
    S["f7"] = _fc(S["f6"], p["f2"], p["fb2"], True)
    return _fc(S["f7"], p["f3"], p["fb3"]), S


def backward(p, x, S, dlogits):
    B, _, N = x.shape
    g = _fc(dlogits, p["f3t"]) * (S["f7"] > 0)
    g = _fc(g, p["f2t"]) * (S["f6"] > 0)
    g = _fc(g, p["f1t"]) * (S["p5"] > 0)
    G128 = _wide_bwd(g, S["i5"], p["w5"], S["h4"], 3)
    G64a = _conv(G128, p["w4t"]) * (S["h3"] > 0)
    dh2p = _conv(G64a, p["w3t"])
    T64 = S["t64"]["T"].view(B, 64, 64)
    gT64 = torch.einsum("bin,bjn->bij", S["h2"], dh2p).reshape(B, 4096)
    dh2 = torch.einsum("bij,bjn->bin", T64, dh2p)
    G = _tnet_bwd(p["t64"], S["t64"], gT64)
    dh2 = (dh2 + _conv(G, p["t64"]["w1"].t())) * (S["h2"] > 0)
    dh1 = _conv(dh2, p["w2t"]) * (S["h1"] > 0)
    T3 = S["t3"]["T"].view(B, 3, 3)
    q = _conv(dh1, p["w1"].t())                                  # d/d x'
    dx = torch.einsum("bdc,bcn->bdn", T3, q)
    gT3 = torch.einsum("bdn,bcn->bdc", x, q).reshape(B, 9)
    G = _tnet_bwd(p["t3"], S["t3"], gT3)
    return dx + _conv(G, p["t3"]["w1"].t())
