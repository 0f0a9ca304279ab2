"""Cell-exact model of the MSB-first interleaved product on the lane-group engine (csrc/mont_msb.hpp).

Every lane's window cell is a 64-bit register with wrap-around; beside it the model keeps the unbounded value of every cell and
asserts that none of a lane's own cells ever wraps, that every quotient digit is the true one or one below, and the product.  Run: python tools/msb_model.py [rounds]
"""
import random
import sys

RB = 29
B = 1 << RB
MASK = B - 1
M64 = (1 << 64) - 1


class Params:
    def __init__(self, M, NLL, T, U):
        self.M, self.NLL, self.T, self.U = M, NLL, T, U
        self.NL = NL = NLL * T
        mtop = (M.bit_length() - 1) // RB
        self.off = NL - 1 - mtop
        assert self.off >= 0
        self.Mt = M << (RB * self.off)
        self.P = self.Mt.bit_length()
        self.tb = self.P - RB * (NL - 1)
        self.ok = 3 <= self.tb <= 26 and self.off >= 1
        W = B ** NL - self.Mt
        self.w = [(W >> (RB * i)) & MASK for i in range(NL)]
        self.mu = (1 << (self.P + 31)) // self.Mt
        assert self.mu < (1 << 32)
        self.norm_blocks = (20 // U)
        assert self.norm_blocks * U <= 20 and NL % U == 0


def limbs(x, n):
    return [(x >> (RB * i)) & MASK for i in range(n)]


def msb_mul(p, a, b, stats=None):
    NLL, T, U, NL = p.NLL, p.T, p.U, p.NL
    NW = NLL + U
    al, bl = limbs(a, NL), limbs(b, NL)
    acc = [[0] * NW for _ in range(T)]
    big = [[0] * NW for _ in range(T)]          # unbounded shadow
    A_true = 0
    since = 0
    NB = NL // U
    for blk in range(NB):
        for u in range(U):
            k = NL - 1 - (blk * U + u)
            idx = k - p.off
            bv = bl[idx] if idx >= 0 else 0
            o = U - 1 - u
            for t in range(T):
                for j in range(NLL):
                    pr = al[t * NLL + j] * bv
                    acc[t][j + o] = (acc[t][j + o] + pr) & M64
                    big[t][j + o] += pr
            A_true = A_true * B + a * bv
            top = acc[T - 1]
            c3, c2, c1, c0 = top[o + NLL], top[o + NLL - 1], top[o + NLL - 2], top[o + NLL - 3]
            x1 = (c1 + 8 * (c0 >> 32)) & M64
            assert c1 + 8 * (c0 >> 32) <= M64, "x1 overflow"
            x2 = x1 >> (p.tb - 3)
            V = ((c2 & 0xFFFFFFFF) * (1 << (32 - p.tb)) + x2) & M64
            vh = ((V >> 32) + ((c2 >> 32) << (32 - p.tb)) + ((c3 & 0xFFFFFFFF) << (RB - p.tb))) & 0xFFFFFFFF
            vl = V & 0xFFFFFFFF
            tt = vh * p.mu + ((vl * p.mu) >> 32)
            q = (tt >> 31) & 0xFFFFFFFF
            if vh >> 31:
                q = 0
            q_true = A_true // p.Mt
            assert q <= q_true, ("overestimate", q, q_true)
            assert q >= q_true - 1, ("underestimate", q, q_true)
            if stats is not None:
                stats[q_true - q] = stats.get(q_true - q, 0) + 1
                stats["qmax"] = max(stats.get("qmax", 0), q)
            for t in range(T):
                for j in range(NLL):
                    pr = p.w[t * NLL + j] * q
                    acc[t][j + o] = (acc[t][j + o] + pr) & M64
                    big[t][j + o] += pr
            A_true -= q * p.Mt
            assert 0 <= A_true < 2 * p.Mt
        last = blk == NB - 1
        # budget: no cell wraps, but the last lane's cells above its own range (aligned columns >= NL: what every reduction's q W leaves
        # there is a multiple of B^NL, read through its low bits only and dropped by the slide)
        for t in range(T):
            for j in range(NW):
                if t == T - 1 and j >= NLL:
                    continue
                assert big[t][j] <= M64, ("cell overflow", blk, t, j)
        # every norm_blocks blocks: carry-save normalisation of the whole window BEFORE the hand-over, the top cell included
        # (its carry is one more value for the next lane): a cell that changes lanes is then as small as one that stays
        since += 1
        carry = [0] * T
        if since == p.norm_blocks and not last:
            since = 0
            for t in range(T):
                for j in range(NW - 1, 0, -1):
                    acc[t][j] = ((acc[t][j] & MASK) + (acc[t][j - 1] >> RB)) & M64 if j < NW - 1 else (acc[t][j] + (acc[t][j - 1] >> RB)) & M64
                acc[t][0] &= MASK
                carry[t] = acc[t][NW - 1] >> RB
                acc[t][NW - 1] &= MASK
                for j in range(NW):
                    big[t][j] = acc[t][j]
        # hand the U cells above the lane's own range to the next lane, then slide the window up by U cells
        for t in range(T - 2, -1, -1):
            for kk in range(U):
                acc[t + 1][kk] = (acc[t + 1][kk] + acc[t][NLL + kk]) & M64
                big[t + 1][kk] += big[t][NLL + kk]
                assert big[t + 1][kk] <= M64 or (t + 1 == T - 1 and kk >= NLL - 1), "handover overflow"
            acc[t + 1][U] = (acc[t + 1][U] + carry[t]) & M64
            big[t + 1][U] += carry[t]
        for t in range(T):
            for j in range(NLL - 1, -1, -1):
                acc[t][j + U] = acc[t][j]
                big[t][j + U] = big[t][j]
            for j in range(U):
                acc[t][j] = 0
                big[t][j] = 0
    # finish (the last block slid too: the own cells are [U, NLL + U)): carries through every lane, the top limb masked
    r = 0
    c = 0
    for t in range(T):
        for j in range(NLL):
            v = acc[t][j + U] + c
            r |= (v & MASK) << (RB * (t * NLL + j))
            c = v >> RB
    assert r == A_true, "finish mismatch"          # (the q B^NL every reduction leaves above the window never reaches the limbs)
    if r >= p.Mt:
        r -= p.Mt
    assert r < p.Mt
    assert r & ((1 << (RB * p.off)) - 1) == 0
    return r >> (RB * p.off)


def rand_modulus(bits, rng):
    while True:
        n = rng.getrandbits(bits // 2) | (1 << (bits // 2 - 1)) | 1
        if n.bit_length() == bits // 2:
            M = n * n
            if M.bit_length() == bits:
                return M


GEOS = {1024: (36, 2, 6), 2048: (36, 4, 6), 3072: (28, 8, 4), 4096: (36, 8, 6)}

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = random.Random(1)
    for key, (NLL, T, U) in GEOS.items():
        stats = {}
        for it in range(rounds):
            M = rand_modulus(2 * key, rng)
            if it == 1:                                   # extreme moduli: top bits all ones / minimal
                M = (1 << (2 * key)) - 1 - 2 * rng.getrandbits(40)
            if it == 2:
                M = (1 << (2 * key - 1)) + 1 + 2 * rng.getrandbits(40)
            p = Params(M, NLL, T, U)
            assert p.ok, (key, p.tb)
            cases = [(M - 1, M - 1), (1, 1), (0, 5), (M - 1, 1), (1, M - 1), (M - 1, (1 << (2 * key - 3)) - 1),
                     (M // 2, M - 2), (rng.getrandbits(64), rng.getrandbits(64))]
            for _ in range(4):
                cases.append((rng.randrange(M), rng.randrange(M)))
            # operands made of all-ones limbs (largest products in every column)
            ones = min(M - 1, (1 << (M.bit_length() - 1)) - 1)
            cases.append((ones, ones))
            # operands that are not reduced (any word pattern of the row: the product must still be the canonical residue)
            full = (1 << (2 * key)) - 1
            cases += [(full, full), (full, M - 1), (rng.getrandbits(2 * key), rng.getrandbits(2 * key))]
            for a, b in cases:
                got = msb_mul(p, a, b, stats)
                assert got == a * b % M, (key, it, "wrong product")
        print(key, "tb", p.tb, "off", p.off, "stats", stats)
    print("ok")
