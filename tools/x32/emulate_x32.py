"""CPU emulation of the index maps of csrc/onf_x32_impl.h (the 32x32x16 ONF kernel).  Tools only, no GPU.

Emulates, lane by lane, the gfx950 semantics the kernel relies on
  * v_mfma_f32_32x32x16_bf16 operand / accumulator maps (cdna_hip_programming.md section 3)
  * ds_read_b128 row reads and ds_read_b64_tr_b16 transposed reads of ONE swizzled LDS image per matrix
  * the accumulator-as-next-B-operand chain (k order permuted inside a 16-block)
and runs the kernel's address formulas (kept literally as in the .hip file) on random weights in float64, comparing
logit and input gradient with a plain numpy MLP.  A pass proves: image packing, both read patterns, the slot <->
position permutation, the folded biases / skip row / ones unit, and the blob (third level) fragment order.
"""
import numpy as np

H = 100
RS1, RS2 = 448, 256
W1_ROWS, W2_ROWS = 103, 101
W1_ZERO, W2_ZERO = 102, 100
SKIP, ONES = 100, 101


def pos_of_slot(s):
    kb, w = s >> 4, s & 15
    g, h, r = (w >> 3) & 1, (w >> 2) & 1, w & 3
    return 16 * kb + 8 * h + 4 * g + r


def swz1(row):
    return (row >> 2) & 3


def swz2(row):
    return ((row & 3) << 2) | ((row >> 2) & 3)


def addr1(row, ch, half=0):
    return row * RS1 + 16 * (ch ^ swz1(row)) + 8 * half


def addr2(row, ch, half=0):
    return row * RS2 + 16 * (ch ^ swz2(row)) + 8 * half


class Net:
    def __init__(self, fin, rng):
        self.fin = fin
        self.W1 = rng.standard_normal((H, fin)); self.b1 = rng.standard_normal(H)
        self.W2 = rng.standard_normal((H, H)); self.b2 = rng.standard_normal(H)
        self.W3 = rng.standard_normal(H + fin); self.b3 = rng.standard_normal()

    def w1ext(self, P, f):
        """row P of the extended first layer at input feature position f"""
        fin = self.fin
        if P < H:
            return self.W1[P, f] if f < fin else (self.b1[P] if f == fin else 0.0)
        if P == SKIP:
            return self.W3[H + f] if f < fin else (self.b3 if f == fin else 0.0)
        if P == ONES:
            return 1.0 if f == fin else 0.0
        return 0.0

    def w2ext(self, P, hp):
        if P >= H:
            return 0.0
        if hp < H:
            return self.W2[P, hp]
        return self.b2[P] if hp == ONES else 0.0


def build_images(net, nslots1=224):
    """element arrays indexed by BYTE address / 2 (one bf16 element = one float64 here)"""
    img1 = np.zeros(W1_ROWS * RS1 // 2)
    for P in range(W1_ROWS):
        for s in range(nslots1):
            img1[(addr1(P, s >> 3) + 2 * (s & 7)) // 2] = net.w1ext(P, pos_of_slot(s))
    img2 = np.zeros(W2_ROWS * RS2 // 2)
    for P in range(W2_ROWS):
        for s in range(112):
            img2[(addr2(P, s >> 3) + 2 * (s & 7)) // 2] = net.w2ext(P, pos_of_slot(s))
    return img1, img2


def read_b128(img, byte_addr):
    assert byte_addr % 16 == 0
    return img[byte_addr // 2: byte_addr // 2 + 8].copy()


def read_tr_b64(img, lane_addr):
    """ds_read_b64_tr_b16 for a whole wave: lane_addr[64] byte addresses -> out[64][4]"""
    out = np.zeros((64, 4))
    for grp in range(4):
        blk = np.zeros((4, 16))
        for q in range(4):
            for p in range(4):
                a = lane_addr[16 * grp + 4 * q + p]
                assert a % 8 == 0
                blk[q, 4 * p: 4 * p + 4] = img[a // 2: a // 2 + 4]
        for i in range(16):
            out[16 * grp + i] = blk[:, i]
    return out


def mfma32(A, B, C):
    """A[64][8], B[64][8] lane fragments, C[64][16] accumulators (f64 here)"""
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        r, h = l & 31, l >> 5
        Am[r, 8 * h: 8 * h + 8] = A[l]
        Bm[8 * h: 8 * h + 8, r] = B[l]
    D = Am @ Bm
    out = C.copy()
    for l in range(64):
        col, g = l & 31, l >> 5
        for reg in range(16):
            out[l, reg] += D[(reg & 3) + 8 * (reg >> 2) + 4 * g, col]
    return out


# ---- the kernel's fragment fetches (address formulas as in onf_x32_impl.h) ---------------------------------------------
def fwd_frag_w1(img1, kb, mt):
    A = np.zeros((64, 8))
    for l in range(64):
        i, g = l & 31, l >> 5
        row = min(32 * mt + i, W1_ZERO)
        x = (i >> 2) & 3
        low = ((2 * (kb & 1) + g) ^ x) & 3
        a = row * RS1 + 64 * (kb >> 1) + 16 * low
        A[l] = read_b128(img1, a)
    return A


def fwd_frag_w2(img2, kb, mt):
    A = np.zeros((64, 8))
    for l in range(64):
        i, g = l & 31, l >> 5
        row = min(32 * mt + i, W2_ZERO)
        rowbase = row * RS2 + 16 * swz2(row)
        a = (rowbase ^ (g << 4)) ^ (kb << 5)
        A[l] = read_b128(img2, a)
    return A


def tr_frag_w2(img2, mt, kb):
    """A'[m' = 32 mt + i][k' slot 16 kb + 8 g + e] for the L2T GEMM: rows of the image are k', chunks are m'"""
    A = np.zeros((64, 8))
    for eh in range(2):
        ad = []
        for l in range(64):
            g, a, q, p = l >> 5, (l >> 4) & 1, (l >> 2) & 3, l & 3
            row = min(16 * kb + 8 * eh + 4 * g + q, W2_ZERO)
            base = row * RS2 + 16 * ((q << 2) | (((2 * a + (p & 1)) ^ (2 * eh + g)) & 3)) + 8 * (p >> 1)
            ad.append(base ^ (mt << 6))
        A[:, 4 * eh: 4 * eh + 4] = read_tr_b64(img2, ad)
    return A


def tr_frag_w1(img1, mt, kb):
    A = np.zeros((64, 8))
    for eh in range(2):
        ad = []
        for l in range(64):
            g, a, q, p = l >> 5, (l >> 4) & 1, (l >> 2) & 3, l & 3
            row = min(16 * kb + 8 * eh + 4 * g + q, W1_ZERO)
            ad.append(row * RS1 + 64 * mt + 16 * (((2 * a + (p & 1)) ^ (2 * eh + g)) & 3) + 8 * (p >> 1))
        A[:, 4 * eh: 4 * eh + 4] = read_tr_b64(img1, ad)
    return A


# ---- blob (third level) fragments: element values in consumption order, checked against the LDS fragments ----------
def blob_frag(net, gemm, kb, mt, l):
    i, g = l & 31, l >> 5
    out = np.zeros(8)
    for e in range(8):
        slot = 16 * kb + 8 * g + e
        if gemm == 'L1':
            out[e] = net.w1ext(32 * mt + i, pos_of_slot(slot))
        elif gemm == 'L2':
            out[e] = net.w2ext(32 * mt + i, pos_of_slot(slot))
        elif gemm == 'L2T':
            out[e] = net.w2ext(pos_of_slot(slot), 32 * mt + i)
        else:
            out[e] = net.w1ext(pos_of_slot(slot), 32 * mt + i)
    return out


def run(fin, seed=0):
    rng = np.random.default_rng(seed)
    net = Net(fin, rng)
    nkb = (fin + 1 + 15) // 16           # ones feature at position fin
    nmt = (nkb + 1) // 2
    img1, img2 = build_images(net)
    x = rng.standard_normal((32, fin))   # 32 points, input features by position
    # reference
    a1 = x @ net.W1.T + net.b1; h1 = np.maximum(a1, 0)
    a2 = h1 @ net.W2.T + net.b2; h2 = np.maximum(a2, 0)
    logit = h2 @ net.W3[:H] + x @ net.W3[H:] + net.b3
    dh2 = net.W3[:H] * (a2 > 0)
    dh1 = (dh2 @ net.W2) * (a1 > 0)
    din = dh1 @ net.W1 + net.W3[H:]

    # ---- L1: B element e of lane (j, g) in block kb = feature position 16 kb + 8 (e >> 2) + 4 g + (e & 3)
    def feat(j, pos):
        return x[j, pos] if pos < fin else (1.0 if pos == fin else 0.0)
    acc1 = [np.zeros((64, 16)) for _ in range(4)]
    for kb in range(nkb):
        B = np.zeros((64, 8))
        for l in range(64):
            j, g = l & 31, l >> 5
            for e in range(8):
                B[l, e] = feat(j, 16 * kb + 8 * (e >> 2) + 4 * g + (e & 3))
        for mt in range(4):
            A = fwd_frag_w1(img1, kb, mt)
            for l in range(64):
                assert np.array_equal(A[l], blob_frag(net, 'L1', kb, mt, l)), ('L1 blob', kb, mt, l)
            acc1[mt] = mfma32(A, B, acc1[mt])
    # check a1 (+bias) and skip row, ones row
    for l in range(64):
        j, g = l & 31, l >> 5
        for mt in range(4):
            for r in range(16):
                P = 32 * mt + 8 * (r >> 2) + 4 * g + (r & 3)
                v = acc1[mt][l, r]
                if P < H: assert abs(v - a1[j, P]) < 1e-9
                elif P == SKIP: assert abs(v - (x[j] @ net.W3[H:] + net.b3)) < 1e-9
                elif P == ONES: assert v == 1.0
                else: assert v == 0.0
    # ---- L2: B block kb = registers 8 (kb & 1) + e of tile kb >> 1 (relu)
    acc2 = [np.zeros((64, 16)) for _ in range(4)]
    for kb in range(7):
        B = np.maximum(acc1[kb >> 1][:, 8 * (kb & 1): 8 * (kb & 1) + 8], 0)
        for mt in range(4):
            A = fwd_frag_w2(img2, kb, mt)
            for l in range(64):
                assert np.array_equal(A[l], blob_frag(net, 'L2', kb, mt, l)), ('L2 blob', kb, mt, l)
            acc2[mt] = mfma32(A, B, acc2[mt])
    lg = np.zeros(64)
    for l in range(64):
        j, g = l & 31, l >> 5
        for mt in range(4):
            for r in range(16):
                P = 32 * mt + 8 * (r >> 2) + 4 * g + (r & 3)
                v = acc2[mt][l, r]
                if P < H:
                    assert abs(v - a2[j, P]) < 1e-9
                    lg[l] += net.W3[P] * max(v, 0)
                else: assert v == 0.0
    for j in range(32):
        got = lg[j] + lg[j + 32] + acc1[3][j + 32, 0]     # skip value: position 100 = tile 3, g = 1, reg 0
        assert abs(got - logit[j]) < 1e-9, (got, logit[j])
    # ---- L2T: dh2 block kb from acc2, output accd in h1 positions
    accd = [np.zeros((64, 16)) for _ in range(4)]
    for kb in range(7):
        B = np.zeros((64, 8))
        for l in range(64):
            g = l >> 5
            for e in range(8):
                P = 16 * kb + 8 * (e >> 2) + 4 * g + (e & 3)
                a = acc2[kb >> 1][l, 8 * (kb & 1) + e]
                B[l, e] = net.W3[P] if (P < H and a > 0) else 0.0
        for mt in range(4):
            A = tr_frag_w2(img2, mt, kb)
            for l in range(64):
                assert np.array_equal(A[l], blob_frag(net, 'L2T', kb, mt, l)), ('L2T blob', kb, mt, l)
            accd[mt] = mfma32(A, B, accd[mt])
    # dh1 = accd * [a1 > 0]; position 100 := 1
    d1 = [np.zeros((64, 16)) for _ in range(4)]
    for l in range(64):
        j, g = l & 31, l >> 5
        for mt in range(4):
            for r in range(16):
                P = 32 * mt + 8 * (r >> 2) + 4 * g + (r & 3)
                v = accd[mt][l, r] if acc1[mt][l, r] > 0 else 0.0
                if P == SKIP: v = 1.0
                d1[mt][l, r] = v
                if P < H: assert abs(v - dh1[j, P]) < 1e-9
    # ---- L1T: output tiles over input positions
    for mt in range(nmt):
        acc = np.zeros((64, 16))
        for kb in range(7):
            B = d1[kb >> 1][:, 8 * (kb & 1): 8 * (kb & 1) + 8]
            A = tr_frag_w1(img1, mt, kb)
            for l in range(64):
                assert np.array_equal(A[l], blob_frag(net, 'L1T', kb, mt, l)), ('L1T blob', kb, mt, l)
            acc = mfma32(A, B, acc)
        for l in range(64):
            j, g = l & 31, l >> 5
            for r in range(16):
                P = 32 * mt + 8 * (r >> 2) + 4 * g + (r & 3)
                if P < fin: assert abs(acc[l, r] - din[j, P]) < 1e-9, (mt, l, r, P)
    print(f'fin={fin}: L1 / L2 / L2T / L1T fragments, chain, folded bias / skip / ones: OK')


if __name__ == '__main__':
    for fin in (220, 200, 120, 100):
        run(fin)
