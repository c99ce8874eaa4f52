"""Host-side RNG contract: raw-word peeking/advancing reproduces numpy's legacy stream and python's
`random` exactly; the BLAS evaluation order assumed by the in-kernel informed sampler is what numpy
does on THIS box (platform-dependent row of SURVEY.md Appendix A)."""
import math
import random

import numpy as np

from nirrt_star_amd import sampling


def test_np_word_stream_matches_random_sample_and_uniform():
    np.random.seed(123)
    w = sampling.peek_np_words(2000)
    d = sampling.words_to_doubles(w)
    got = np.array([np.random.random_sample() for _ in range(500)])
    assert np.array_equal(got, d[:500])
    u = np.array([np.random.uniform(3, 221) for _ in range(500)])
    assert np.array_equal(u, 3 + (221 - 3) * d[500:1000])
    # peeking did not consume; advancing does
    np.random.seed(123)
    sampling.peek_np_words(64)
    a = np.random.random_sample()
    np.random.seed(123)
    sampling.advance_np_words(10)
    b = np.random.random_sample()
    assert a == d[0] and b == d[5]


def test_py_word_stream_matches_random_uniform():
    random.seed(77)
    w = sampling.peek_py_words(1000)
    d = sampling.words_to_doubles(w)
    got = np.array([random.uniform(-1, 1) for _ in range(300)])
    assert np.array_equal(got, -1.0 + 2.0 * d[:300])
    random.seed(77)
    sampling.advance_py_words(8)
    assert random.random() == d[4]


def test_blas_forms_of_informed_sampling_on_this_box():
    rng = np.random.default_rng(0)
    bad2 = bad3 = 0
    for _ in range(2000):
        th = rng.uniform(-3, 3)
        C = np.array([[math.cos(th), -math.sin(th), 0], [math.sin(th), math.cos(th), 0], [0, 0, 1.0]])
        r = np.array([rng.uniform(50, 150), rng.uniform(5, 60)])
        r = np.array([r[0], r[1], r[1]])
        x, y = rng.uniform(-1, 1, 2)
        xb = np.array([[x], [y], [0.0]])
        CL = np.dot(C, np.diag(r))
        assert np.array_equal(CL, C * r[None, :])
        res = np.dot(CL, xb)
        for i in range(2):
            bad2 += math.fma(CL[i, 0], x, CL[i, 1] * y) != res[i, 0] if hasattr(math, "fma") else _fma(CL[i, 0], x, CL[i, 1] * y) != res[i, 0]
        A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        CL3 = A @ np.diag(r)
        v = rng.uniform(-1, 1, 3)
        res3 = CL3 @ v
        for i in range(3):
            bad3 += _fma(CL3[i, 2], v[2], _fma(CL3[i, 0], v[0], CL3[i, 1] * v[1])) != res3[i]
    assert bad2 == 0 and bad3 == 0


def _fma(a, b, c):
    from fractions import Fraction
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def test_randint_is_masked_rejection_on_single_words():
    """np.random.randint(0, m) of the legacy generator == mask-and-reject on 32-bit outputs (what the kernel's
    SamplePointCloud emulates)"""
    np.random.seed(5)
    w = sampling.peek_np_words(40000)
    pos = 0
    for m in [1, 2, 3, 7, 8, 9, 700, 1023, 1024, 1025, 2048] * 40:
        exp = np.random.randint(0, m)
        rng = m - 1
        got = 0
        if rng:
            mask = rng
            for sft in (1, 2, 4, 8, 16):
                mask |= mask >> sft
            while True:
                got = int(w[pos]) & mask
                pos += 1
                if got <= rng:
                    break
        assert got == exp
    # interleaved with doubles like the NIRRT* sampler does
    assert np.random.random() == sampling.words_to_doubles(w[pos:pos + 2])[0]
