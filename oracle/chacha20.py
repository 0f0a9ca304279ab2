"""TEST INFRASTRUCTURE (oracle): ChaCha20 block function, restated from RFC 8439 sections 2.1-2.3 (a published
algorithm; there is no reference source for it in /root/reference — upstream ipcl draws obfuscator randomness
from IPP-Crypto's generators, which are not in the tree).  Pinned by the RFC's own known-answer vectors
(sections 2.1.1 and 2.3.2) in tests/test_oracle.py.  Only tests may import this module.

`draw_r_words` restates the row layout of `pai_draw_r` (include/paillier_hip.h): word w of the key stream is
word w of the [N][r_words] matrix, top word of every row masked to randbits; the 32-bit block counter carries
into nonce word 0."""
import numpy as np

MASK = 0xFFFFFFFF


def _rotl(v: int, c: int) -> int:
    return ((v << c) & MASK) | (v >> (32 - c))


def quarter_round(a: int, b: int, c: int, d: int):
    """RFC 8439 section 2.1."""
    a = (a + b) & MASK; d ^= a; d = _rotl(d, 16)
    c = (c + d) & MASK; b ^= c; b = _rotl(b, 12)
    a = (a + b) & MASK; d ^= a; d = _rotl(d, 8)
    c = (c + d) & MASK; b ^= c; b = _rotl(b, 7)
    return a, b, c, d


def block(key_words, counter: int, nonce_words):
    """RFC 8439 section 2.3: 16 output words of one block."""
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter & MASK] + list(nonce_words)
    x = list(s)
    for _ in range(10):
        for (i, j, k, l) in ((0, 4, 8, 12), (1, 5, 9, 13), (2, 6, 10, 14), (3, 7, 11, 15),
                             (0, 5, 10, 15), (1, 6, 11, 12), (2, 7, 8, 13), (3, 4, 9, 14)):
            x[i], x[j], x[k], x[l] = quarter_round(x[i], x[j], x[k], x[l])
    return [(a + b) & MASK for a, b in zip(x, s)]


def draw_r_words(key: bytes, nonce: bytes, counter0: int, n_rows: int, r_words: int, randbits: int) -> np.ndarray:
    kw = np.frombuffer(key, dtype="<u4").tolist()
    nw = np.frombuffer(nonce, dtype="<u4").tolist()
    total = n_rows * r_words
    out = []
    blk = 0
    while len(out) < total:
        ctr = counter0 + (blk & MASK)
        n0 = (nw[0] + (blk >> 32)) & MASK
        out.extend(block(kw, ctr, [n0, nw[1], nw[2]]))
        blk += 1
    a = np.array(out[:total], dtype=np.uint32).reshape(n_rows, r_words)
    top = randbits - 32 * (r_words - 1)
    if top < 32:
        a[:, -1] &= np.uint32((1 << top) - 1)
    return a
