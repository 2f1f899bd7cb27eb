"""Property tests of the CPU oracle itself (hypothesis, CPU only): invariants any correct restatement must have,
independent of the golden vectors — pack/unpack inverses, quantiser ranges and error bounds, the Kronecker identity,
Hadamard orthogonality, the integer GEMM against numpy's, the paged-cache append against a dense history."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import fq_oracle as O

SET = dict(max_examples=25, deadline=None)


@settings(**SET)
@given(st.integers(1, 9), st.integers(1, 16), st.integers(0, 2**31 - 1))
def test_pack_unpack_inverse(rows, half_cols, seed):
    q = np.random.default_rng(seed).integers(-8, 8, (rows, 2 * half_cols)).astype(np.int32)
    p = O.pack_i4(q)
    assert p.dtype == np.uint8 and p.shape == (rows, half_cols)
    assert np.array_equal(O.unpack_i4(p), q)
    assert np.array_equal(p & 15, q[:, 0::2] & 15) and np.array_equal(p >> 4, q[:, 1::2] & 15)   # even column = low nibble


@settings(**SET)
@given(st.integers(1, 6), st.sampled_from([8, 32, 100]), st.floats(0.05, 1.0), st.floats(0.05, 1.0), st.booleans(),
       st.integers(0, 2**31 - 1))
def test_symmetric_quantiser_ranges_and_error(rows, cols, smax, smin, clamp0, seed):
    rng = np.random.default_rng(seed)
    y = (rng.standard_normal((rows, cols)) * rng.uniform(0.01, 50, (rows, 1))).astype(np.float32)
    o = O.quant_outputs(y, smax, smin, clamp0=clamp0)
    q = O.unpack_i4(o["packed"])
    assert q.min() >= -8 and q.max() <= 7
    s = o["scale16"].astype(np.float32).reshape(rows, 1)
    inside = (y <= 7 * s) & (y >= -8 * s)                       # un-clipped values: at most half a step (+ fp16 scale rounding)
    err = np.abs(q * o["scale"].reshape(rows, 1) - y)
    assert np.all(err[inside] <= 0.5 * o["scale"].reshape(rows, 1).repeat(cols, 1)[inside] * 1.001 + 1e-6)
    assert np.all(o["scale"] > 0)


@settings(**SET)
@given(st.sampled_from([(2, 4), (3, 8), (8, 8), (5, 6)]), st.integers(1, 4), st.integers(0, 2**31 - 1))
def test_kronecker_identity(shape, rows, seed):
    M, N = shape
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((rows, M * N)).astype(np.float16)
    L = (rng.standard_normal((M, M)) / np.sqrt(M)).astype(np.float16)
    R = (rng.standard_normal((N, N)) / np.sqrt(N)).astype(np.float16)
    y = O.kron_transform(x, L, R).reshape(rows, -1)
    ref = x.astype(np.float64) @ np.kron(L.astype(np.float64), R.astype(np.float64))
    assert np.max(np.abs(y - ref)) <= 4e-3 * (np.abs(ref).max() + 1)   # fp16 rounding of the intermediate U


@settings(**SET)
@given(st.sampled_from([8, 64, 512]), st.integers(1, 3), st.integers(0, 2**31 - 1))
def test_hadamard_is_an_orthogonal_involution(n, rows, seed):
    x = np.random.default_rng(seed).standard_normal((rows, n)).astype(np.float16)
    y = O.hadamard(x, 1, None)
    assert np.allclose(np.linalg.norm(y.astype(np.float64), axis=1), np.linalg.norm(x.astype(np.float64), axis=1), rtol=3e-3)
    z = O.hadamard(y, 1, None)
    assert np.max(np.abs(z.astype(np.float32) - x.astype(np.float32))) <= 1e-2 * max(1.0, np.abs(x).max())


@settings(**SET)
@given(st.integers(1, 9), st.integers(1, 9), st.sampled_from([32, 64, 96]), st.integers(0, 2**31 - 1))
def test_int4_matmul_is_the_integer_product(M, N, K, seed):
    rng = np.random.default_rng(seed)
    xq, wq = rng.integers(-8, 8, (M, K)).astype(np.int32), rng.integers(-8, 8, (N, K)).astype(np.int32)
    assert np.array_equal(O.int4_matmul(O.pack_i4(xq), O.pack_i4(wq)), (xq.astype(np.int64) @ wq.T.astype(np.int64)).astype(np.int32))


@settings(**SET)
@given(st.booleans(), st.integers(1, 5), st.sampled_from([16, 64, 128]), st.integers(0, 2**31 - 1))
def test_kv_asymmetric_quantiser(lac, rows, hd, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, hd)) * rng.uniform(0.05, 8, (rows, 1))).astype(np.float16)
    p, s, z, q = O.kv_asym_quant(x, np.float16(0.97), np.float16(0.93), lac)
    assert q.min() >= 0 and q.max() <= 15 and np.array_equal(p, q[:, 0::2] | (q[:, 1::2] << 4))
    if not lac:   # no clipping: every value within half a step (+ the fp16 roundings of the chain)
        d = O.kv_asym_dequant(p, s, z, lac).astype(np.float32)
        assert np.all(np.abs(d - x.astype(np.float32)) <= 0.56 * s.astype(np.float32) + 2e-3)


@settings(**SET)
@given(st.integers(1, 3), st.sampled_from([4, 16]), st.integers(1, 20), st.integers(0, 2**31 - 1))
def test_paged_append_reconstructs_the_history(batch, page_size, length, seed):
    """append `length` tokens one by one (decode path) and all at once (prefill path): same cache contents, and reading the
    pages back in order gives the tokens in order."""
    rng = np.random.default_rng(seed)
    heads, hd, layers = 2, 16, 2
    n_pg = (length + page_size - 1) // page_size
    indices = rng.permutation(batch * n_pg).astype(np.int32)
    k = rng.integers(0, 256, (batch, length, heads, hd // 2), dtype=np.uint8)
    v = rng.integers(0, 256, (batch, length, heads, hd // 2), dtype=np.uint8)
    kp = rng.uniform(0.1, 1, (batch, length, heads, 2)).astype(np.float16)
    vp = rng.uniform(0.1, 1, (batch, length, heads, 2)).astype(np.float16)

    def fresh():
        return (np.zeros((batch * n_pg, layers, 2, heads, page_size, hd // 2), np.uint8),
                np.zeros((batch * n_pg, layers, 2, heads, page_size, 2), np.float16))

    def specs(n):
        pg = (n + page_size - 1) // page_size
        ind = np.arange(batch + 1, dtype=np.int32) * pg
        idx = np.concatenate([indices[b * n_pg:b * n_pg + pg] for b in range(batch)])
        return ind, idx, np.full(batch, (n - 1) % page_size + 1, np.int32)

    d1, p1 = fresh()
    ind, idx, last = specs(length)
    O.kv_cache_append(d1, p1, ind, idx, last, 1, k.reshape(-1, heads, hd // 2), v.reshape(-1, heads, hd // 2),
                      kp.reshape(-1, heads, 2), vp.reshape(-1, heads, 2), np.arange(batch + 1, dtype=np.int32) * length)
    d2, p2 = fresh()
    for t in range(length):
        ind, idx, last = specs(t + 1)
        O.kv_cache_append(d2, p2, ind, idx, last, 1, k[:, t], v[:, t], kp[:, t], vp[:, t])
    assert np.array_equal(d1, d2) and np.array_equal(p1.view(np.uint16), p2.view(np.uint16))
    ind, idx, last = specs(length)
    for b in range(batch):
        rows = [d1[idx[ind[b] + pos // page_size], 1, 0, :, pos % page_size] for pos in range(length)]
        assert np.array_equal(np.stack(rows), k[b])
    assert not d1[:, 0].any()                                   # the other layer is untouched
