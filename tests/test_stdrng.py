"""`StdRng::seed_from_u64(5)` (scenes.rs:64) — rand 0.8.3's StdRng is ChaCha12 behind a PCG32-expanded
seed.  The crates are not in /root/reference and cannot be built here, so:

  * the ChaCha block function of both implementations (oracle/src/stdrng.h and csrc/host/std_rng.hpp, written
    separately) is pinned by RFC 7539's published vectors: the quarter round of section 2.1.1 and the 20-round block of
    section 2.3.2;
  * seeding (PCG32-XSH-RR, rand_core 0.6 `seed_from_u64`), word order and the f64 conversions (`Standard`,
    `UniformFloat::sample_single`) are restated from the crates' published algorithms and have NO vector available
    offline: PARITY UNPINNED.  What is checked is that the two implementations agree with each other and with a third
    restatement in this file, and the statistical sanity of the streams.
Runs without a GPU."""
import ctypes as C
import struct

import numpy as np

RFC_STATE = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574, 0x03020100, 0x07060504, 0x0B0A0908, 0x0F0E0D0C,
             0x13121110, 0x17161514, 0x1B1A1918, 0x1F1E1D1C, 0x00000001, 0x09000000, 0x4A000000, 0x00000000]
RFC_BLOCK = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
             0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]
RFC_QUARTER_IN = [0x11111111, 0x01020304, 0x9B8D6F43, 0x01234567]
RFC_QUARTER_OUT = [0xEA2A92F4, 0xCB1CF8CE, 0x4581472E, 0x5881C4BB]
M32 = 0xFFFFFFFF


def _impls():
    import oracle_ffi as O
    from raytracer_amd import _ffi as F
    orc = O.lib()
    return (("oracle", orc.orc_chacha_block, orc.orc_chacha_quarter, orc.orc_stdrng_draws),
            ("host", F.host.rth_kat_chacha, F.host.rth_kat_chacha_quarter, F.host.rth_kat_stdrng))


def test_chacha_core_matches_rfc7539():
    for name, block, quarter, _ in _impls():
        q = (C.c_uint32 * 4)(*RFC_QUARTER_IN)
        quarter(q)
        assert list(q) == RFC_QUARTER_OUT, name
        st, out = (C.c_uint32 * 16)(*RFC_STATE), (C.c_uint32 * 16)()
        block(st, 20, out)
        assert list(out) == RFC_BLOCK, name


# ---- a third restatement (pure Python) of the seeding and conversions, for cross-checking ----------------------------
def _rotl(v, n):
    return ((v << n) | (v >> (32 - n))) & M32


def _block12(state):
    x = list(state)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 7)
    for _ in range(6):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M32 for a, b in zip(x, state)]


def _python_stdrng_words(seed, n):
    s, key = seed, []
    for _ in range(8):
        s = (s * 6364136223846793005 + 11634580027462260723) & 0xFFFFFFFFFFFFFFFF
        xs = (((s >> 18) ^ s) >> 27) & M32
        rot = s >> 59
        key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & M32)
    state = RFC_STATE[:4] + key + [0, 0, 0, 0]
    words = []
    while len(words) < 2 * n:
        words += _block12(state)
        state[12] = (state[12] + 1) & M32
        if state[12] == 0:
            state[13] = (state[13] + 1) & M32
    return [words[2 * i] | (words[2 * i + 1] << 32) for i in range(n)]


def test_both_implementations_and_the_python_restatement_agree():
    n = 600  # 600 u64 = 75 blocks: crosses many block boundaries
    for seed in (5, 0, 1, 0xFFFFFFFFFFFFFFFF, 123456789012345):
        want = _python_stdrng_words(seed, n)
        got = {}
        for name, _, _, draws in _impls():
            w = np.zeros(n, dtype=np.uint64)
            u = np.zeros(n, dtype=np.float64)
            r = np.zeros(n, dtype=np.float64)
            draws.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
            draws(seed, n, w.ctypes.data, u.ctypes.data, r.ctypes.data, 0.5, 1.0)
            assert [int(x) for x in w] == want, (name, seed)
            # Standard: top 53 bits / 2^53; gen_range: top 52 bits as a [1, 2) mantissa
            assert np.array_equal(u, np.array([(x >> 11) * 2.0 ** -53 for x in want]))
            m = [struct.unpack("<d", struct.pack("<Q", (x >> 12) | 0x3FF0000000000000))[0] for x in want]
            assert np.array_equal(r, np.array([(v - 1.0) * 0.5 + 0.5 for v in m]))
            assert (u >= 0).all() and (u < 1).all() and (r >= 0.5).all() and (r < 1.0).all()
            got[name] = (w, u, r)
        assert all(np.array_equal(a, b) for a, b in zip(got["oracle"], got["host"]))


def test_stream_statistics():
    from raytracer_amd import _ffi as F
    n = 200000
    w, u, r = np.zeros(n, dtype=np.uint64), np.zeros(n), np.zeros(n)
    F.host.rth_kat_stdrng(5, n, w.ctypes.data, u.ctypes.data, r.ctypes.data, 0.0, 0.5)
    assert abs(u.mean() - 0.5) < 4 * (1 / 12 / n) ** 0.5 and abs(u.var() - 1 / 12) < 1e-3
    assert abs(r.mean() - 0.25) < 4 * (0.25 / 12 / n) ** 0.5
    bits = np.unpackbits(w.view(np.uint8))
    assert abs(bits.mean() - 0.5) < 4 * 0.5 / (bits.size ** 0.5)


def test_scene_14_is_random_scene_with_other_numbers():
    """same construction as scene 0 (scenes.rs:56-122): a ground sphere, <= 484 small spheres on the 22 x 22 grid, three
    big ones; the host layer and the oracle build it from their own generators and must agree to the bit
    (test_host_bvh.py covers the tree; here: counts and the seeded quantities)."""
    import raytracer_amd as R
    a, b = R.HostScene(14, 16 / 9), R.HostScene(0, 16 / 9)
    da, db = a.desc, b.desc
    assert 400 < da.n_spheres + da.n_mspheres <= 488 and da.n_mspheres > 250
    sa = np.ctypeslib.as_array(da.msphere, shape=(da.n_mspheres * 9,)).reshape(-1, 9)
    assert (sa[:, 1] == 0.2).all() and (sa[:, 4] >= 0.2).all() and (sa[:, 4] < 0.7).all() and (sa[:, 6] == 0.2).all()
    sb = np.ctypeslib.as_array(db.msphere, shape=(db.n_mspheres * 9,)).reshape(-1, 9)
    assert not np.array_equal(np.sort(sa[:, 0])[:50], np.sort(sb[:, 0])[:50])  # different generators, different scenes
    again = R.HostScene(14, 16 / 9)
    assert np.array_equal(sa, np.ctypeslib.as_array(again.desc.msphere, shape=(da.n_mspheres * 9,)).reshape(-1, 9))
