"""CPU emulation of fd_conv_wino.hip's index arithmetic (tile enumeration over levels / images / dilation parity classes,
4x4 patch gather, B^T d B, G g G^T, per-frequency GEMM, A^T M A, scatter) against F.conv2d.  Development aid; the GPU tests
(tests/test_layers_gpu.py::test_conv3x3_winograd*) are the parity check of the kernel itself."""
import numpy as np
import torch
import torch.nn.functional as F

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def tile_table(batch, hw, dil):
    """per level: TH, TW (tile grid of ONE parity class), first tile"""
    th = [-(-(-(-h // dil)) // 2) for h, _ in hw]
    tw = [-(-(-(-w // dil)) // 2) for _, w in hw]
    t0 = [0]
    for i in range(len(hw)):
        t0.append(t0[-1] + batch * dil * dil * th[i] * tw[i])
    return th, tw, t0


def decode(t, batch, hw, dil, th, tw, t0):
    s = 0
    for i in range(1, len(hw)):
        if t >= t0[i]:
            s = i
    local = t - t0[s]
    cls_tiles = th[s] * tw[s]
    tpi = dil * dil * cls_tiles
    n, r = divmod(local, tpi)
    cls, r2 = divmod(r, cls_tiles)
    ti, tj = divmod(r2, tw[s])
    ca, cb = divmod(cls, dil)
    return s, n, ca + 2 * dil * ti, cb + 2 * dil * tj


def wino_conv(xs, w, dil):
    """xs: list of [B, C, H, W] levels; w [Cout, Cin, 3, 3]"""
    B = xs[0].shape[0]
    hw = [tuple(x.shape[2:]) for x in xs]
    th, tw, t0 = tile_table(B, hw, dil)
    U = np.einsum("ir,ocrq,jq->ijoc", G, w.double().numpy(), G)           # [4,4,Cout,Cin]
    ys = [np.full((B, w.shape[0], h, ww), np.nan) for h, ww in hw]
    for t in range(t0[-1]):
        s, n, h0, w0 = decode(t, B, hw, dil, th, tw, t0)
        H, W = hw[s]
        d = np.zeros((4, 4, w.shape[1]))
        for i in range(4):
            for j in range(4):
                hh, ww = h0 + (i - 1) * dil, w0 + (j - 1) * dil
                if 0 <= hh < H and 0 <= ww < W:
                    d[i, j] = xs[s][n, :, hh, ww].double().numpy()
        V = np.einsum("ia,abc,jb->ijc", BT, d, BT)
        M = np.einsum("ijoc,ijc->ijo", U, V)
        Y = np.einsum("xi,ijo,yj->xyo", AT, M, AT)
        for x in range(2):
            for y in range(2):
                h, ww = h0 + x * dil, w0 + y * dil
                if h < H and ww < W:
                    assert np.isnan(ys[s][n, 0, h, ww]), "pixel written twice"
                    ys[s][n, :, h, ww] = Y[x, y]
    return ys


if __name__ == "__main__":
    g = torch.Generator().manual_seed(0)
    for dil, hw in ((1, [(5, 5), (3, 4), (1, 1)]), (2, [(8, 8), (5, 7), (2, 3)]), (1, [(6, 10)]), (2, [(4, 4)])):
        xs = [torch.randn(2, 8, h, w, generator=g) for h, w in hw]
        w = torch.randn(5, 8, 3, 3, generator=g)
        ys = wino_conv(xs, w, dil)
        for x, y in zip(xs, ys):
            ref = F.conv2d(x.double(), w.double(), None, 1, dil, dil).numpy()
            assert not np.isnan(y).any(), "pixel never written"
            print(dil, tuple(x.shape[2:]), "max err", np.abs(y - ref).max())
            assert np.abs(y - ref).max() < 1e-10
    print("ok")
