"""CPU: index-for-index Python model of the HIP NTT pass structure (csrc/kernels_ntt.hpp ntt_pass_kernel,
zkr_prove.hip ntt_plan) and of the calcH pipeline constants, checked against the oracle NTT / calcH."""
import random

import groth16 as g
from bn254 import R


def bitrev(x, L):
    return int(bin(x)[2:].zfill(L)[::-1], 2) if L else 0


def make_tw(m):
    w = g.root_of_unity(2 * m)
    t = [1] * m
    for k in range(1, m):
        t[k] = t[k - 1] * w % R
    return t


def tw_lookup(table, tlog, slog, x, inverse):
    idx = x << (tlog - slog)
    if not inverse:
        return table[idx]
    return 1 if idx == 0 else (-table[(1 << tlog) - idx]) % R


def ntt_plan(L, tile_log, strided_log):
    v = []
    lows = min(L, tile_log)
    rem = L - lows
    if rem > 0:
        nch = (rem + strided_log - 1) // strided_log
        hi = L
        for i in range(nch):
            nb = (rem + (nch - i) - 1) // (nch - i)
            lo = hi - nb
            v.append((lo, hi, min(tile_log - nb, lo)))
            hi = lo
            rem -= nb
    v.append((0, lows, 0))
    return v


def ntt_pass(x, L, lo, hi, wlog, dif, inverse, tw, tlog, twl, twl_log, fused=True):
    nb, W = hi - lo, 1 << wlog
    rows = 1 << nb
    tile = rows << wlog
    lb = (1 << lo) >> wlog
    for blk in range((1 << L) // tile):
        q, l0 = blk // lb, (blk % lb) << wlog
        base = (q << hi) + l0
        lds = [x[base + ((e >> wlog) << lo) + (e & (W - 1))] for e in range(tile)]
        def twiddle(slog, imods):
            table, tl = (twl, twl_log) if slog <= twl_log else (tw, tlog)
            return tw_lookup(table, tl, slog, imods, inverse)

        def bf(u, v, w):
            if dif:
                return (u + v) % R, (u - v) * w % R
            v = v * w % R
            return (u + v) % R, (u - v) % R

        def single(j):
            sl = 1 << j
            for b in range(tile >> 1):
                c, kk = b & (W - 1), b >> wlog
                r0 = ((kk >> j) << (j + 1)) | (kk & (sl - 1))
                e0 = (r0 << wlog) + c
                e1 = e0 + (sl << wlog)
                w = twiddle(j + lo, ((kk & (sl - 1)) << lo) + l0 + c)
                lds[e0], lds[e1] = bf(lds[e0], lds[e1], w)

        def double(j):  # stages j and j+1 on four elements held in registers (one LDS round trip, one barrier)
            sl = 1 << j
            for t in range(tile >> 2):
                c, kq = t & (W - 1), t >> wlog
                xlow = kq & (sl - 1)
                r00 = ((kq >> j) << (j + 2)) | xlow
                e = [((r00 + a * sl + b * 2 * sl) << wlog) + c for b in (0, 1) for a in (0, 1)]   # e00, e01, e10, e11
                wa = twiddle(j + lo, (xlow << lo) + l0 + c)
                wb0 = twiddle(j + 1 + lo, (xlow << lo) + l0 + c)
                wb1 = twiddle(j + 1 + lo, ((xlow + sl) << lo) + l0 + c)
                x00, x01, x10, x11 = (lds[i] for i in e)
                if dif:     # stage j+1 first
                    a0, a2 = bf(x00, x10, wb0)
                    a1, a3 = bf(x01, x11, wb1)
                    y00, y01 = bf(a0, a1, wa)
                    y10, y11 = bf(a2, a3, wa)
                else:       # stage j first
                    a0, a1 = bf(x00, x01, wa)
                    a2, a3 = bf(x10, x11, wa)
                    y00, y10 = bf(a0, a2, wb0)
                    y01, y11 = bf(a1, a3, wb1)
                lds[e[0]], lds[e[1]], lds[e[2]], lds[e[3]] = y00, y01, y10, y11

        if not fused:
            for jj in range(nb):
                single(nb - 1 - jj if dif else jj)
        elif dif:
            j = nb - 1
            if nb & 1:
                single(j)
                j -= 1
            while j >= 1:
                double(j - 1)
                j -= 2
        else:
            j = 0
            if nb & 1:
                single(0)
                j = 1
            while j + 1 < nb:
                double(j)
                j += 2
        for e in range(tile):
            x[base + ((e >> wlog) << lo) + (e & (W - 1))] = lds[e]


def run_ntt(x, L, dif, inverse, tw, tlog, twl, twl_log, tile_log=4, strided_log=3, fused=True):
    plan = ntt_plan(L, tile_log, strided_log)
    if not dif:
        plan = plan[::-1]
    for lo, hi, wlog in plan:
        ntt_pass(x, L, lo, hi, wlog, dif, inverse, tw, tlog, twl, twl_log, fused)


def test_pass_structure_matches_oracle_ntt():
    rnd = random.Random(5)
    twl_log = 3                      # scaled-down local table (device: 10), tile 2^4 (device: 2^11)
    twl = make_tw(1 << twl_log)
    for L in (1, 3, 4, 6, 9, 11):
        m = 1 << L
        tw = make_tw(m)
        v = [rnd.randrange(R) for _ in range(m)]
        ref = g.ntt(v)
        x = list(v)
        for fused in (True, False):       # the device kernel fuses pairs of stages; the plain form is the same network
            x = list(v)
            run_ntt(x, L, True, False, tw, L, twl, twl_log, fused=fused)
            assert [x[bitrev(i, L)] for i in range(m)] == ref, L
            y = [ref[bitrev(i, L)] for i in range(m)]
            run_ntt(y, L, False, True, tw, L, twl, twl_log, fused=fused)
            minv = pow(m, R - 2, R)
            assert [a * minv % R for a in y] == v, L
        # tile / strided geometries with odd and even stage counts per pass
        for tile_log, strided_log in ((5, 3), (3, 2), (6, 4)):
            x = list(v)
            run_ntt(x, L, True, False, tw, L, twl, twl_log, tile_log, strided_log)
            assert [x[bitrev(i, L)] for i in range(m)] == ref, (L, tile_log)


def test_device_plan_shapes():
    for L in range(1, 28):
        plan = ntt_plan(L, 11, 9)
        assert plan[-1] == (0, min(L, 11), 0)
        bits = sorted((lo, hi) for lo, hi, _ in plan)
        assert bits[0][0] == 0 and bits[-1][1] == L and all(a[1] == b[0] for a, b in zip(bits, bits[1:]))
        for lo, hi, wlog in plan[:-1]:
            assert hi - lo <= 9 and hi - lo + wlog <= 11 and wlog <= lo and lo >= 11


def test_calc_h_pipeline_constants():
    """the GPU route: 2 iNTT(DIF) -> coset scale by tw[bitrev] -> 2 NTT(DIT) -> 2 iNTT(DIF) of products -> combine"""
    circ = g.synth_circuit(64, 5, 0x5A4B0001)
    sc = g.setup_scalars(circ, g.toxic_from_seed(1))
    pk = dict(nVars=circ["nVars"], nPublic=5, domainSize=64, polsA=sc["polsA"], polsB=sc["polsB"])
    m, L = 64, 6
    tw, twl = make_tw(m), make_tw(8)
    MONT = (1 << 256) % R
    Rinv = pow(MONT, R - 2, R)
    montmul = lambda a, b: a * b * Rinv % R
    for w in (circ["witness"], [(x + (i == 10)) % R for i, x in enumerate(circ["witness"])]):
        a, b = g.qap_evaluate(pk, w)
        A, B = list(a), list(b)
        run_ntt(A, L, True, True, tw, L, twl, 3)
        run_ntt(B, L, True, True, tw, L, twl, 3)
        A = [A[p] * tw[bitrev(p, L)] % R for p in range(m)]
        B = [B[p] * tw[bitrev(p, L)] % R for p in range(m)]
        run_ntt(A, L, False, False, tw, L, twl, 3)
        run_ntt(B, L, False, False, tw, L, twl, 3)
        U = [montmul(x, y) for x, y in zip(A, B)]
        V = [montmul(x, y) for x, y in zip(a, b)]
        run_ntt(U, L, True, True, tw, L, twl, 3)
        run_ntt(V, L, True, True, tw, L, twl, 3)
        minv, half = pow(m, R - 2, R), pow(2, R - 2, R)
        c1 = MONT * MONT % R * half % R * minv % R
        c2 = c1 * minv % R * minv % R
        h_br = []
        for pos in range(m):
            i = bitrev(pos, L)
            ginv_mont = MONT if i == 0 else (-tw[m - i] * MONT) % R
            h_br.append((montmul(V[pos], c1) - montmul(montmul(U[pos], ginv_mont), c2)) % R)
        assert [h_br[bitrev(i, L)] for i in range(m)] == g.calc_h_websnark(pk, w)


def test_value_bounds_of_the_lazily_reduced_passes():
    """Model of the bounds the 29-bit-limb butterflies keep (csrc/kernels_ntt.hpp, NTT_H_*), in half moduli.  DIT: a read at
    stage j states NTT_H_DIT; the true bound after j stages is NTT_H_IN + 4 j (a product is below 2 r: bound 3 or 4, a
    subtraction of it adds 2 r), so the statement holds as long as no pass has more than NTT_TILE_LOG stages, and the
    operand of every product stays inside what field29.hpp's mul admits (Ha * Hb <= 676, twiddle bound 2).  DIF: every
    value written back is at most NTT_H_DIF because the one double sum of a pair of stages goes through barrett() (bound 5)."""
    H_IN, H_DIF, TILE_LOG, STRIDED_LOG = 6, 6, 11, 9
    H_DIT = H_IN + 4 * TILE_LOG
    for L in range(1, 28):
        for (lo, hi, wlog) in ntt_plan(L, TILE_LOG, STRIDED_LOG):
            nb = hi - lo
            assert 1 <= nb <= TILE_LOG
            h = H_IN
            for j in range(nb):                    # DIT stages: t = v w (bound 3), u + t and u - t + 2 r
                assert h <= H_DIT and h * 2 <= 676
                h += 4
            assert h <= H_DIT + 4 <= 128           # what barrett() takes at the end of the pass
    # DIF pair of stages on four values of bound H: (s0, p0), (s1, p1) = dif(x00, x10), dif(x01, x11); then dif(s0, s1), dif(p0, p1)
    def dif(hu, hv):
        hsum, hdiff = hu + hv, hu + 2 * ((hv + 1) // 2 + 1)     # round 3: the difference is a sweep-less factor (sub_loose: one more p)
        assert hdiff * 2 <= 676 and (hv + 1) // 2 + 1 <= 24
        return hsum, 3
    s0, p0 = dif(H_DIF, H_DIF)
    s1, p1 = dif(H_DIF, H_DIF)
    ss, sp = dif(s0, s1)
    ps, pp = dif(p0, p1)
    assert ss <= 128 and max(5, sp, ps, pp) <= H_DIF   # ss goes through barrett() -> 5
    s, p = dif(H_DIF, H_DIF)                           # a single stage
    assert s <= 128 and max(5, p) <= H_DIF
