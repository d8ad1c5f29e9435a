"""Oracle: numpy restatement of the in-kernel noise generator (Philox4x32-10 + Box-Muller).

The reference has no counterpart: it draws noise from torch's global generator
(/root/reference/src/sampling.py:13,54,74,79).  The product's throughput mode replaces that with a
counter-based generator (didi_amd/csrc/elementwise.hip); this file restates the published
Philox4x32-10 algorithm (Salmon et al., SC'11; Random123 constants) so tests can check the raw
32-bit stream bit-for-bit and the normals to float tolerance.  The Philox round function is
additionally pinned by the Random123 known-answer vectors in tests/test_oracle_philox.py.
TEST INFRASTRUCTURE: see oracle/__init__.py.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr4, key2):
    """ctr4: uint32 [n, 4], key2: uint32 [2] -> uint32 [n, 4]."""
    c = ctr4.astype(np.uint64)
    k0, k1 = int(key2[0]), int(key2[1])
    for _ in range(10):
        p0 = M0 * c[:, 0]
        p1 = M1 * c[:, 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[:, 1] ^ np.uint64(k0)) & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[:, 3] ^ np.uint64(k1)) & MASK
        c = np.stack([n0, p1 & MASK, n2, p0 & MASK], axis=1)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def raw_stream(n4, seed, offset):
    """The kernel's counter layout: counter = (offset + i) as 64 bits in words 0,1; words 2,3 = 0."""
    i = np.arange(n4, dtype=np.uint64) + np.uint64(offset)
    ctr = np.zeros((n4, 4), dtype=np.uint32)
    ctr[:, 0] = (i & MASK).astype(np.uint32)
    ctr[:, 1] = (i >> np.uint64(32)).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return philox4x32_10(ctr, key)


def normal(n, seed, offset=0):
    """float32 [n] (n % 4 == 0): element 4i+j comes from counter offset+i."""
    assert n % 4 == 0
    r = raw_stream(n // 4, seed, offset).astype(np.float32)
    k = np.float32(2.3283064365386963e-10)
    out = np.empty((n // 4, 4), dtype=np.float32)
    for p in range(2):
        u1 = (r[:, 2 * p] + np.float32(0.5)) * k
        u2 = (r[:, 2 * p + 1] + np.float32(0.5)) * k
        rad = np.sqrt(np.float32(-2.0) * np.log(u1))
        ang = np.float32(6.283185307179586) * u2
        out[:, 2 * p] = rad * np.cos(ang)
        out[:, 2 * p + 1] = rad * np.sin(ang)
    return out.reshape(-1)
