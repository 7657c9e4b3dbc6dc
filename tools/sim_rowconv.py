#!/usr/bin/env python3
"""CPU model of the address arithmetic of csrc/conv_row.h (development aid, no GPU needed): the DMA placement, the
swizzled fragment reads (must return the logical (row, x, slot) the MFMA step expects, zero pixels for the padding) and
the LDS bank conflicts of every ds_read_b128 of the MFMA loop."""
import sys

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in grp] for grp in GROUPS]


def run(C):
    W, P = 2048 // C, 2 * C
    S, KS, TH = P // 16, C // 16, 8
    PXP, ROWB, NROWS = 1024 // P, (W + 1) * P, TH + 2
    g = (lambda x: (x >> 2) & 3) if C == 32 else (lambda x: (x >> 1) & 7)
    lds = {}          # 16-byte slot index -> content
    for k in range(NROWS + 1):
        base = 0 if k == 0 else P + (k - 1) * ROWB + W * P
        for j in range(S):
            lds[(base + j * 16) // 16] = "zero"
    for rs in range(NROWS):
        for wave in range(4):
            for lane in range(64):
                px = lane // S
                xcol = wave * PXP + px
                lslot = (lane % S) ^ g(xcol)
                # source byte offset inside the row: wave*1024 + px*P + lslot*16 -> pixel / slot
                off = wave * 1024 + px * P + lslot * 16
                assert off // P == xcol and (off % P) // 16 == lslot
                dst = P + rs * ROWB + wave * 1024 + lane * 16
                assert dst % 16 == 0 and dst // 16 not in lds, "overlap"
                lds[dst // 16] = (rs, xcol, lslot)
    assert len(lds) == (P + NROWS * ROWB) // 16, (len(lds), (P + NROWS * ROWB) // 16)
    worst = 1
    for wave in range(4):
        sel, rg = wave & 1, wave >> 1
        col = sel if W == 64 else 0
        for irel in range(6):
            for dxi in range(3):
                for kk in range(KS):
                    addrs = []
                    for lane in range(64):
                        l31, half = lane & 31, lane >> 5
                        xq = col * 32 + l31 + dxi - 1
                        a = P + rg * 4 * ROWB + xq * P + (((2 * kk + half) ^ g(xq)) << 4) + irel * ROWB
                        assert a % 16 == 0 and a >= 0
                        got = lds[a // 16]
                        if xq < 0 or xq >= W:
                            assert got == "zero", (wave, irel, dxi, kk, lane, got)
                        else:
                            assert got == (rg * 4 + irel, xq, 2 * kk + half), (wave, irel, dxi, kk, lane, got)
                        addrs.append(a)
                    for grp in GROUPS:
                        banks = {}
                        for l in grp:
                            b = (addrs[l] // 16) % 16       # 16-byte slot of the 256-byte bank row
                            banks.setdefault(b, set()).add(addrs[l])
                        worst = max(worst, max(len(v) for v in banks.values()))
    print(f"C={C}: placement and reads consistent; worst ds_read_b128 conflict {worst}-way; tile bytes {P + NROWS * ROWB}")


for C in (32, 64):
    run(C)


def run_deep(C, y0=0):
    W, P = 2048 // C, 2 * C
    S, KS, TH = P // 16, C // 16, 8
    PXP, ROWB, NROWS = 1024 // P, (W + 1) * P, TH + 2
    MW = 1 if C == 128 else 2
    NW, RPT = 4 // MW, 32 // W
    f = lambda y, x: (y * W + x) & 15
    lds = {}
    for k in range(NROWS + 1):
        base = 0 if k == 0 else P + (k - 1) * ROWB + W * P
        for j in range(S):
            lds[(base + j * 16) // 16] = "zero"
    for rs in range(NROWS):
        y = y0 - 1 + rs
        for wave in range(4):
            for lane in range(64):
                px, pslot = lane // S, lane % S
                xcol = wave * PXP + px
                lslot = pslot ^ f(y, xcol)
                off = wave * 1024 + px * P + lslot * 16
                assert off // P == xcol and (off % P) // 16 == lslot and lslot < S
                dst = P + rs * ROWB + wave * 1024 + lane * 16
                assert dst // 16 not in lds
                lds[dst // 16] = (rs, xcol, lslot)
    assert len(lds) == (P + NROWS * ROWB) // 16
    worst = 1
    ncls = 6 if C == 256 else 3
    for lane_half in range(1):
        for kk in range(KS):
            for tap in range(9):
                dyi, dxi = tap // 3, tap % 3
                for t in range(NW):
                    addrs = []
                    for lane in range(64):
                        l31, half = lane & 31, lane >> 5
                        r, x = l31 // W, l31 % W
                        yr = t * RPT + r
                        corner = P + yr * ROWB + (x - 1) * P
                        cls = ((dyi + 1) & 1) * 3 + dxi if C == 256 else dxi
                        par = cls // 3
                        fy = ((y0 + yr + par) & 1) if C == 256 else 0
                        bt = corner ^ (((half ^ f(fy, x + (cls % 3) - 1)) & (S - 1)) << 4)
                        a = (bt ^ (kk << 5)) + dyi * ROWB + dxi * P
                        got = lds[a // 16]
                        xs, rs = x + dxi - 1, yr + dyi       # source column, LDS row slot
                        if xs < 0 or xs >= W:
                            assert got == "zero", (C, kk, tap, t, lane, got)
                        else:
                            assert got == (rs, xs, 2 * kk + half), (C, kk, tap, t, lane, got, (rs, xs, 2 * kk + half))
                        addrs.append(a)
                    for grp in GROUPS:
                        banks = {}
                        for l in grp:
                            banks.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
                        worst = max(worst, max(len(v) for v in banks.values()))
    print(f"deep C={C} y0={y0}: placement and reads consistent; worst ds_read_b128 conflict {worst}-way; tile bytes {P + NROWS * ROWB}")


for C in (128, 256):
    run_deep(C, 0)
    run_deep(C, 8)


def run_img(C):
    W, P = 2048 // C, 2 * C
    S, KS = P // 16, C // 16
    NT = 8 if C == 128 else 4
    TPI = W * W // 32
    T, PPT, PXP = 33 * P, 32 * P // 1024, 1024 // P
    lds = {}
    for t in range(NT):
        for j in range(S):
            lds[(t * T + 32 * P + j * 16) // 16] = "zero"
    for wave in range(4):
        for i in range(16):
            q = wave + 4 * i
            tq, pq = q // PPT, q % PPT
            for lane in range(64):
                px, pslot = lane // S, lane % S
                pp = pq * PXP + px                      # pixel index inside the tile
                lslot = pslot ^ (pp & 15)
                off = q * 1024 + px * P + lslot * 16    # source offset inside the 64 KiB group
                gpix = off // P                         # global pixel of the group
                assert gpix == tq * 32 + pp and (off % P) // 16 == lslot
                dst = tq * T + pq * 1024 + lane * 16
                assert dst // 16 not in lds
                lds[dst // 16] = (tq, pp, lslot)
    assert len(lds) == NT * T // 16
    worst = 1
    RPT = 32 // W
    for kk in range(KS):
        for tap in range(9):
            dyi, dxi = tap // 3, tap % 3
            for t in range(NT):
                addrs = []
                for lane in range(64):
                    l31, half = lane & 31, lane >> 5
                    r, x = l31 // W, l31 % W

                    def bconst(dyi, dxi, row_ok):
                        xs = x + dxi - 1
                        ps = l31 + (dyi - 1) * W + dxi - 1
                        rel = ps * P + (-P if ps < 0 else P if ps >= 32 else 0)
                        v = rel ^ (((half ^ (ps & 15)) & (S - 1)) << 4)
                        return v if (row_ok and 0 <= xs < W) else 32 * P
                    if dyi == 0 and t % TPI == 0:
                        b = bconst(0, dxi, r > 0)
                    elif dyi == 2 and t % TPI == TPI - 1:
                        b = bconst(2, dxi, r < RPT - 1)
                    else:
                        b = bconst(dyi, dxi, True)
                    a = (b ^ (kk << 5)) + t * T
                    got = lds[a // 16]
                    # expected: image coordinates
                    img, ti = t // TPI, t % TPI
                    y = ti * RPT + r
                    ys, xs = y + dyi - 1, x + dxi - 1
                    if ys < 0 or ys >= W or xs < 0 or xs >= W:
                        assert got == "zero", (C, kk, tap, t, lane, got)
                    else:
                        gp = img * W * W + ys * W + xs        # pixel of the group
                        assert got == (gp // 32, gp % 32, 2 * kk + half), (C, kk, tap, t, lane, got, (gp // 32, gp % 32, 2 * kk + half))
                    addrs.append(a)
                for grp in GROUPS:
                    banks = {}
                    for l in grp:
                        banks.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
                    worst = max(worst, max(len(v) for v in banks.values()))
    print(f"img C={C}: placement and reads consistent; worst ds_read_b128 conflict {worst}-way; tile bytes {NT * T}")


for C in (128, 256):
    run_img(C)
