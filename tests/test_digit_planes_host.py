"""Host-side (numpy) restatement of the fixed-point digit scheme of csrc/gram_i8.h / snapshot_i8.h / nn_gemm_i8.h: the arithmetic facts the int8 kernels rest on,
checked without a GPU — digit ranges, exact recombination of the nine digit products, what dropping the weight-4 product costs, the int32 bound of a row segment,
and the exponent rule (max |x| 2^-E < 127/128).  The kernels themselves are checked against int64 arithmetic in tests/test_gpu_gram_i8.py."""
import numpy as np


def exponent_of_max(mx):
    """E of colmaxexp_kernel / gi_exp_of_key: mx = f 2^e2, f in [0.5, 1); one more when f >= 127/128"""
    f, e2 = np.frexp(np.float32(mx))
    return int(e2) + int(f >= np.float32(127.0 / 128.0))


def digits(t):
    """balanced radix-256 digits of split_i8_kernel: t = d0 2^16 + d1 2^8 + d2 (numpy int64 in, three int64 arrays out)"""
    t = np.asarray(t, dtype=np.int64)
    d2 = ((t & 0xff) ^ 0x80) - 0x80          # sign-extended low byte
    t1 = (t - d2) >> 8
    d1 = ((t1 & 0xff) ^ 0x80) - 0x80
    d0 = (t1 - d1) >> 8
    return d0, d1, d2


def test_digit_ranges_and_reconstruction_over_the_whole_24_bit_range():
    lim = 127 * 65536
    t = np.concatenate([np.arange(-lim, -lim + 70000), np.arange(-70000, 70000), np.arange(lim - 70000, lim + 1),
                        np.random.default_rng(0).integers(-lim, lim + 1, 200000)])
    d0, d1, d2 = digits(t)
    assert (d0 * 65536 + d1 * 256 + d2 == t).all()
    assert d0.min() >= -127 and d0.max() <= 127            # the top digit fits int8 BECAUSE the scale keeps |t| <= 127 * 2^16
    assert d1.min() >= -128 and d1.max() <= 127 and d2.min() >= -128 and d2.max() <= 127


def test_exponent_rule_keeps_the_scaled_maximum_below_127_128():
    rng = np.random.default_rng(1)
    mx = np.concatenate([np.float32(2.0) ** rng.integers(-120, 120, 2000) * rng.uniform(0.5, 1.0, 2000).astype(np.float32),
                         np.array([1.0, 127.0 / 128.0, np.nextafter(np.float32(127.0 / 128.0), np.float32(0)), 1.9999999, 1e-40], dtype=np.float32)])
    for m in mx:
        E = exponent_of_max(m)
        v = np.ldexp(np.float64(m), -E)
        assert v < 127.0 / 128.0 and v >= 127.0 / 256.0 - 1e-12, (m, E, v)   # below the limit, and never more than one bit wasted
        t = int(np.rint(np.ldexp(np.float64(m), 23 - E)))
        assert abs(t) <= 127 * 65536


def test_nine_digit_products_recombine_exactly_and_the_weight_4_product_is_2_to_the_minus_32():
    rng = np.random.default_rng(2)
    lim = 127 * 65536
    a = rng.integers(-lim, lim + 1, (512, 7))
    b = rng.integers(-lim, lim + 1, (512, 5))
    da, db = digits(a), digits(b)
    P = [np.zeros((7, 5), dtype=np.int64) for _ in range(5)]
    for i in range(3):
        for j in range(3):
            P[i + j] += da[i].T @ db[j]                     # what one int32 accumulator per weight holds
    full = sum(P[s] << (8 * (4 - s)) for s in range(5))
    assert (full == a.T @ b).all()                          # gram_i8_kernel: exact
    eight = sum(P[s] << (8 * (4 - s)) for s in range(4))    # snapshot_i8 / nn_gemm_i8: the weight-4 product dropped
    assert np.abs(full - eight).max() <= 512 * 128 * 128    # <= rows * 2^14, against products of up to 2^46 per row: 2^-32
    six = sum(P[s] << (8 * (4 - s)) for s in range(3))      # the first versions: weight <= 2
    small = rng.integers(-40000, 40000, (512, 7))           # values that live in the two low digits
    ds = digits(small)
    Ps = [np.zeros((7, 7), dtype=np.int64) for _ in range(5)]
    for i in range(3):
        for j in range(3):
            Ps[i + j] += ds[i].T @ ds[j]
    exact = small.T @ small
    rel6 = np.abs(sum(Ps[s] << (8 * (4 - s)) for s in range(3)) - exact).max() / np.abs(exact).max()
    rel8 = np.abs(sum(Ps[s] << (8 * (4 - s)) for s in range(4)) - exact).max() / np.abs(exact).max()
    assert rel6 > 1e-4 and rel8 < rel6 / 30, (rel6, rel8)   # why six products were wrong for entries far below their column's largest (measured here: 4.9e-4 against 1.0e-5 of the largest entry)
    assert np.abs(full - six).max() > np.abs(full - eight).max()


def test_int32_accumulators_hold_a_segment_of_32768_rows():
    # worst case per weight: three products of magnitude 128 * 128 per row (weight 2), all of one sign
    assert 3 * 128 * 128 * 32768 < 2 ** 31
    assert 3 * 128 * 128 * 43691 >= 2 ** 31                 # the bound the drivers enforce with 32768-row segments / the m_pad <= 32768 guards
