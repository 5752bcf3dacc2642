"""Search LDS row strides / chunk swizzles for the 32x32x16 weight images (tools; CPU only).

One bf16 image M[row = output position][slot] serves
  * forward A fragments: ds_read_b128, lane (i = l & 31, g = l >> 5) reads row 32 mt + i, 16-byte chunk 2 kb + g
  * transposed A fragments: ds_read_b64_tr_b16, lane l: g = l >> 5, a = (l >> 4) & 1, q = (l >> 2) & 3, p = l & 3 reads
    row 16 kb + 8 eh + 4 g + q, chunk 2 (2 mt + a) + (p & 1), half p >> 1
Bank rules (MI355X_MICROARCH.md, LDS): b128 is served in four 16-lane groups, tr_b16 in the two 32-lane halves; bank =
(addr / 4) % 64.  Prints the conflict-free (stride, swizzle) pairs.
"""
import itertools, sys

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               [32 + x for x in (list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)))],
               [32 + x for x in (list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)))]]


def cycles(addr_bytes, width):
    """LDS cycles of one lane group: max over banks of distinct dword addresses on that bank."""
    per_bank = {}
    for a in addr_bytes:
        for d in range(width // 4):
            dw = a // 4 + d
            per_bank.setdefault(dw % 64, set()).add(dw)
    return max(len(s) for s in per_bank.values())


def check(RS, f, nrows, nslots, rowmap=lambda r: r):
    nch = nslots // 8
    worst_f = worst_t = 1
    for mt in range((nrows + 31) // 32):
        for kb in range(nslots // 16):
            for grp in B128_GROUPS:
                ad = []
                for l in grp:
                    i, g = l & 31, l >> 5
                    row = rowmap(32 * mt + i)
                    ad.append(row * RS + 16 * ((2 * kb + g) ^ f(row)))
                worst_f = max(worst_f, cycles(ad, 16))
    # transposed: rows = k' (output units, nrows of them), columns = m' positions (nslots of them)
    for mt in range(nslots // 32 + (1 if nslots % 32 else 0)):
        for kb in range((nrows + 15) // 16):
            for eh in range(2):
                for half in range(2):
                    ad = []
                    for l in range(32 * half, 32 * half + 32):
                        g, a, q, p = l >> 5, (l >> 4) & 1, (l >> 2) & 3, l & 3
                        row = rowmap(16 * kb + 8 * eh + 4 * g + q)
                        ch = 2 * (2 * mt + a) + (p & 1)
                        if ch >= nch:
                            ch = nch - 1   # clamp (pad)
                        ad.append(row * RS + 16 * (ch ^ f(row)) + 8 * (p >> 1))
                    worst_t = max(worst_t, cycles(ad, 8))
    return worst_f, worst_t


fams = {
    'none': lambda r: 0,
    'r&3': lambda r: r & 3,
    '(r>>2)&3': lambda r: (r >> 2) & 3,
    '(r&3)<<2|(r>>2)&3': lambda r: ((r & 3) << 2) | ((r >> 2) & 3),
    '(r&3)<<1': lambda r: (r & 3) << 1,
    '(r&3)<<2': lambda r: (r & 3) << 2,
    '(r&1)<<1|(r>>1)&1': lambda r: ((r & 1) << 1) | ((r >> 1) & 1),
    '(r>>1)&3': lambda r: (r >> 1) & 3,
    '(r&7)': lambda r: r & 7,
    '(r&15)': lambda r: r & 15,
    '(r&3)<<2|(r>>2)&1': lambda r: ((r & 3) << 2) | ((r >> 2) & 1),
    '((r&3)^((r>>2)&3))': lambda r: ((r & 3) ^ ((r >> 2) & 3)),
}

for name, (nrows, nslots) in {'W1': (128, 224), 'W2': (128, 112)}.items():
    print(name)
    for RS in range(nslots * 2, nslots * 2 + 144, 16):
        for fn, f in fams.items():
            # the swizzle must keep chunks inside the padded row
            maxch = max((c ^ f(r)) for r in range(16) for c in range(nslots // 8))
            if 16 * (maxch + 1) > RS:
                continue
            wf, wt = check(RS, f, nrows, nslots)
            if wf * wt <= 2:
                print(f'  RS={RS} swz={fn}: fwd {wf}x  tr {wt}x')
