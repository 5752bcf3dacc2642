#!/usr/bin/env python3
"""Index-level emulation of the fused ONF kernel's MFMA data flow (design aid, CPU only).

Emulates v_mfma_f32_16x16x4_f32 with the documented gfx950 lane maps
  A: lane l holds A[l&15][l>>4]   B: lane l holds B[l>>4][l&15]   D: reg r of lane l = D[4*(l>>4)+r][l&15]
and replays the kernel's layer chain (weights = A operand read from a row-major padded LDS image,
activations = B operand taken straight from the previous layer's accumulators) to check that the
feature permutation phi() makes every layer compute the right matrix product, and that every LDS read of
the weight images is bank-conflict-free for ds_read_b32 (32 banks, two 32-lane halves).
"""
import numpy as np

H = 100
LANES = np.arange(64)
I = LANES & 15
G = LANES >> 4


class Layout(object):
    """feature held in D row rho = 4g + r of tile t:  base(t) + a[g] + r."""

    def __init__(self, kind):
        self.kind = kind
        self.a = np.array((0, 16, 4, 20) if kind == "P" else (0, 8, 4, 12))

    def base(self, t):
        return 32 * (t >> 1) + 8 * (t & 1) if self.kind == "P" else 16 * t

    def rowpos(self, i):
        return self.a[i >> 2] + (i & 3)

    def col(self, g):
        return self.a[g]


LF, LH1, LH2 = Layout("P"), Layout("Q"), Layout("P")


def phi(lay, t, rho, hidden):
    g, r = rho >> 2, rho & 3
    if hidden and t == 6:
        return 96 + g if r == 0 else -1
    return lay.base(t) + lay.a[g] + r


def mfma(a, b, c):
    """a,b: [64] lane values, c: [64,4] -> c + A@B in D layout."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    A[I, G] = a
    B[G, I] = b
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[:, r] += D[4 * G + r, I]
    return out


def banks_ok(addr):
    for half in (slice(0, 32), slice(32, 64)):
        a = addr[half]
        bank = a % 32
        for b in np.unique(bank):
            if len(np.unique(a[bank == b])) > 1:
                return False
    return True


def run(fin, nkt, seed=0):
    rng = np.random.default_rng(seed)
    S1 = 225 if nkt > 8 else 129
    S2 = 129
    W1 = rng.normal(size=(H, fin))
    W2 = rng.normal(size=(H, H))
    X = rng.normal(size=(fin, 16))          # features x points (one 16-point tile)
    L1 = np.zeros((H, S1)); L1[:, :fin] = W1
    L2 = np.zeros((H, S2)); L2[:, :H] = W2
    conflicts = {"L1": 0, "L2": 0, "L2T": 0, "L1T": 0}
    reads = dict(conflicts)

    def rd(img, S, row, col, tag):
        addr = row * S + col
        reads[tag] += 1
        if not banks_ok(addr):
            conflicts[tag] += 1
        return img.reshape(-1)[addr]

    def row_m(lay, mt):  # A-lane row for output tile mt (hidden)
        return np.where(mt < 6, lay.base(mt) + lay.rowpos(I), 96 + (I >> 2))

    def kcol(lay, t, r):  # k index of lane group g at k-step (t, r) (hidden layouts)
        return lay.base(t) + r + lay.col(G) if t < 6 else 96 + G

    # ---- L1 forward: a1 = W1 @ X
    acc1 = [np.zeros((64, 4)) for _ in range(7)]
    for ks in range(4 * nkt):
        f = LF.base(ks >> 2) + (ks & 3) + LF.col(G)
        b = np.where(f < fin, X[np.minimum(f, fin - 1), I], 0.0)
        for mt in range(7):
            a = rd(L1, S1, row_m(LH1, mt), f, "L1")
            acc1[mt] = mfma(a, b, acc1[mt])
    ref1 = W1 @ X
    for mt in range(7):
        for r in range(4):
            for l in range(64):
                h = phi(LH1, mt, 4 * G[l] + r, True)
                if h >= 0:
                    assert abs(acc1[mt][l, r] - ref1[h, I[l]]) < 1e-9
    h1 = [np.maximum(a, 0) for a in acc1]

    # ---- L2 forward: a2 = W2 @ h1 ; k-steps (t, r), tile 6 only r = 0
    ksteps = [(t, r) for t in range(6) for r in range(4)] + [(6, 0)]
    acc2 = [np.zeros((64, 4)) for _ in range(7)]
    for (t, r) in ksteps:
        col = kcol(LH1, t, r)
        for mt in range(7):
            a = rd(L2, S2, row_m(LH2, mt), col, "L2")
            acc2[mt] = mfma(a, h1[t][:, r], acc2[mt])
    ref2 = W2 @ np.maximum(ref1, 0)
    for mt in range(7):
        for r in range(4):
            for l in range(64):
                h = phi(LH2, mt, 4 * G[l] + r, True)
                if h >= 0:
                    assert abs(acc2[mt][l, r] - ref2[h, I[l]]) < 1e-9

    # ---- L2^T backward: dh1 = W2^T @ dh2  (dh2 in h2's layout)
    dh2_ref = rng.normal(size=(H, 16))
    dh2 = [np.zeros((64, 4)) for _ in range(7)]
    for mt in range(7):
        for r in range(4):
            for l in range(64):
                h = phi(LH2, mt, 4 * G[l] + r, True)
                dh2[mt][l, r] = dh2_ref[h, I[l]] if h >= 0 else 123.0   # garbage in pads
    accd = [np.zeros((64, 4)) for _ in range(7)]
    for (t, r) in ksteps:
        row = kcol(LH2, t, r)
        for mt in range(7):
            col = row_m(LH1, mt)
            a = rd(L2, S2, row, col, "L2T")
            accd[mt] = mfma(a, dh2[t][:, r], accd[mt])
    refd = W2.T @ dh2_ref
    for mt in range(7):
        for r in range(4):
            for l in range(64):
                h = phi(LH1, mt, 4 * G[l] + r, True)
                if h >= 0:
                    assert abs(accd[mt][l, r] - refd[h, I[l]]) < 1e-9

    # ---- L1^T backward: din = W1^T @ dh1 (dh1 in h1 layout), output in the input-feature layout
    dh1 = accd
    for mt in range(nkt):
        acc = np.zeros((64, 4))
        for (t, r) in ksteps:
            row = kcol(LH1, t, r)
            col = LF.base(mt) + LF.rowpos(I)
            a = rd(L1, S1, row, col, "L1T")
            acc = mfma(a, dh1[t][:, r], acc)
        refi = W1.T @ refd
        for r in range(4):
            f = LF.base(mt) + LF.col(G) + r      # feature of D row (g, r) == slot (ks=4mt+r, g)
            for l in range(64):
                if f[l] < fin:
                    assert abs(acc[l, r] - refi[f[l], I[l]]) < 1e-9
                else:
                    assert abs(acc[l, r]) < 1e-12
    return reads, conflicts


if __name__ == "__main__":
    for fin, nkt in ((220, 14), (100, 7), (200, 13), (120, 8)):
        reads, conf = run(fin, nkt)
        print("FIN=%d NKT=%d: layer chain OK; LDS reads %s; conflicted reads %s" % (fin, nkt, reads, conf))
