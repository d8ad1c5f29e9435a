"""Pin the Philox oracle with the Random123 known-answer vectors (philox4x32-10)."""
import numpy as np

from oracle.philox import normal, philox4x32_10, raw_stream

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def test_random123_known_answers():
    for ctr, key, want in KAT:
        got = philox4x32_10(np.array([ctr], dtype=np.uint32), np.array(key, dtype=np.uint32))[0]
        assert tuple(int(x) for x in got) == want


def test_stream_is_counter_based():
    a = raw_stream(64, seed=0x1234_5678_9ABC, offset=100)
    b = raw_stream(16, seed=0x1234_5678_9ABC, offset=132)
    assert np.array_equal(a[32:48], b)


def test_normal_moments():
    z = normal(1 << 18, seed=7)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert np.isfinite(z).all()
