"""Integer identities the extraction kernels rely on, checked exhaustively on the CPU (numpy): they replace a division, an unpacking loop or a
compare-and-shift sequence in detect.hip and must hold for every input the kernels can see -- the GPU parity tests only sample them."""
import numpy as np


def test_grid_cell_by_multiply_high_equals_the_division():
    # k_fast_select: (v * 2^level) / cell as umulhi(v, ceil(2^32 / cell)); v < 2^16 (level coordinates x scale stay below the image width),
    # every cell size a grid can have (detect.hip: FastArgs::cell_magic)
    v = np.arange(1 << 16, dtype=np.uint64)
    for cell in list(range(2, 130)) + [255, 256, 1000, 4096, 65535]:
        magic = ((1 << 32) + cell - 1) // cell
        assert magic < (1 << 32)
        assert np.array_equal((v * np.uint64(magic)) >> np.uint64(32), v // np.uint64(cell)), cell


def test_ring_masks_by_sign_bits_equal_the_comparisons():
    # k_fast_select pass 1b: bit k of the bright mask = (v_k > p + t) taken as the sign of (p + t - v_k), shifted in from the right (the mask comes out
    # mirrored); the ten-contiguous-bits test must not see the mirroring
    rng = np.random.default_rng(0)

    def ring10(m):
        m = m | (m << 16)
        a = m & (m >> 1); b = a & (a >> 2); c = b & (b >> 4)
        return (c & (a >> 8) & 0xFFFF) != 0

    for _ in range(2000):
        p, t = int(rng.integers(0, 256)), int(rng.integers(1, 60))
        v = rng.integers(0, 256, 16)
        if rng.random() < 0.5:                                  # arcs of equal sign are the interesting inputs
            s, n = int(rng.integers(0, 16)), int(rng.integers(8, 13))
            v[(s + np.arange(n)) % 16] = min(255, p + t + 1)
        bright = sum(int(v[k] > p + t) << k for k in range(16))
        acc = 0
        for k in range(16):
            acc = ((acc << 1) | (((p + t - int(v[k])) >> 31) & 1)) & 0xFFFFFFFF      # v_alignbit(acc, hi - v, 31)
        mirrored = sum(((bright >> k) & 1) << (15 - k) for k in range(16))
        assert acc == mirrored
        assert ring10(acc) == ring10(bright)


def test_intensity_centroid_moments_as_byte_dot_products():
    # k_describe: lane = (row v, half of the row); table entry = |u| of the columns inside the circle (and ones); m10 = +- dot(bytes, |u|),
    # m01 = v * dot(bytes, ones) -- against the definition (FeatureDetector.cpp:509-537)
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    tab_w = np.zeros((64, 16), np.int64); tab_1 = np.zeros((64, 16), np.int64)
    for l in range(62):
        v, hh = (l >> 1) - 15, l & 1
        for k in range(16):
            au = 1 + k if hh else 15 - k
            if au <= umax[abs(v)] and au <= 15:
                tab_w[l, k] = au; tab_1[l, k] = 1
    rng = np.random.default_rng(1)
    for _ in range(200):
        P = rng.integers(0, 256, (39, 40)).astype(np.int64)
        m10 = m01 = 0
        for v in range(-15, 16):
            for u in range(-umax[abs(v)], umax[abs(v)] + 1):
                m10 += u * P[v + 19, u + 19]; m01 += v * P[v + 19, u + 19]
        g10 = g01 = 0
        for l in range(64):
            v, hh = (l >> 1) - 15, l & 1
            row = P[v + 19, 4 + 16 * hh: 20 + 16 * hh]            # column 4 (u = -15) or 20 (u = 1)
            sw, s1 = int((row * tab_w[l]).sum()), int((row * tab_1[l]).sum())
            g10 += sw if hh else -sw; g01 += v * s1
        assert (g10, g01) == (m10, m01)
