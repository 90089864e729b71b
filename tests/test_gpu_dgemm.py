"""FP64 MFMA GEMM building block vs numpy float64 (GPU).  Tolerance: f64
accumulation in a different order, |err| <= 1e-12 * sum|a||b| per element."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

dp = C.POINTER(C.c_double)


def _gemm(A, B, Cm, alpha, beta, b_nmajor, lower_only=0, tile=-1):
    from glmmrmcml_amd import _lib
    L = _lib.lib()
    M, K = A.shape
    N = Cm.shape[1]
    A = np.asfortranarray(A); B = np.asfortranarray(B); out = np.asfortranarray(Cm.copy())
    rc = L.glmmr_mcml_dbg_dgemm(M, N, K, A.ctypes.data_as(dp), A.shape[0], B.ctypes.data_as(dp),
                                B.shape[0], int(b_nmajor), C.c_double(alpha), C.c_double(beta),
                                out.ctypes.data_as(dp), out.shape[0], lower_only, tile)
    _lib.check(rc)
    return out


@pytest.mark.parametrize("M,N,K", [(16, 16, 4), (160, 128, 16), (161, 129, 17), (333, 77, 250),
                                   (1000, 256, 999), (5, 3, 2), (640, 512, 640)])
@pytest.mark.parametrize("b_nmajor", [0, 1])
@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_dgemm_matches_numpy(M, N, K, b_nmajor, tile):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.normal(size=(M, K)); Bm = rng.normal(size=(K, N)); C0 = rng.normal(size=(M, N))
    Bdev = Bm.T.copy() if b_nmajor else Bm
    got = _gemm(A, Bdev, C0, -1.0, 1.0, b_nmajor, tile=tile)
    want = C0 - A @ Bm
    bound = 1e-12 * (np.abs(A) @ np.abs(Bm) + np.abs(C0))
    assert np.all(np.abs(got - want) <= bound)


def test_dgemm_asymmetric_identity_check():
    """A = I with an asymmetric B catches a transposed C write"""
    n = 48
    B = np.arange(n * n, dtype=float).reshape(n, n)
    got = _gemm(np.eye(n), B, np.zeros((n, n)), 1.0, 0.0, 0)
    assert np.array_equal(got, B)


def test_dgemm_lower_only_leaves_upper_tiles():
    rng = np.random.default_rng(3)
    n, k = 700, 128
    A = rng.normal(size=(n, k)); C0 = rng.normal(size=(n, n))
    got = _gemm(A, A, C0, -1.0, 1.0, 1, lower_only=1)
    want = C0 - A @ A.T
    low = np.tril_indices(n)
    assert np.allclose(got[low], want[low], rtol=0, atol=1e-10)
    # tiles strictly above the diagonal are untouched
    assert np.array_equal(got[0:128, 640:], C0[0:128, 640:])


# the sampler's dense direct-to-LDS kernel (dgemm_dlds.h::dgemm_dlds_asm_kernel, the path of a dense non-identity Z with
# more than 16 chains): force_tile 40, operands padded to the contract the sampler meets
@pytest.mark.parametrize("M,N,K", [(160, 128, 32), (161, 129, 33), (333, 77, 250), (1000, 256, 999), (80, 17, 16),
                                   (5000, 1024, 5000)])
def test_dgemm_dlds_matches_numpy(M, N, K):
    rng = np.random.default_rng(M + 3 * N + 7 * K)
    A = rng.normal(size=(M, K)); Bm = rng.normal(size=(K, N)); C0 = rng.normal(size=(M, N))
    got = _gemm(A, Bm, C0, -1.0, 1.0, 0, tile=40)
    want = C0 - A @ Bm
    bound = 1e-12 * (np.abs(A) @ np.abs(Bm) + np.abs(C0))
    assert np.all(np.abs(got - want) <= bound)
    got = _gemm(A, Bm, C0, 2.0, 0.0, 0, tile=40)
    assert np.all(np.abs(got - 2.0 * (A @ Bm)) <= 2 * bound)


def test_dgemm_dlds_asymmetric_identity_check():
    n = 160
    B = np.arange(n * 96, dtype=float).reshape(n, 96)
    got = _gemm(np.eye(n), B, np.zeros((n, 96)), 1.0, 0.0, 0, tile=40)
    assert np.array_equal(got, B)


# the short-K direct-to-LDS kernel of the Cholesky / TRSM panel updates (dgemm_dl.h): K % 16 == 0
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (4872, 128, 128), (1000, 1024, 128), (131, 77, 16),
                                   (5, 3, 32), (777, 300, 256), (64, 128, 128)])
@pytest.mark.parametrize("b_nmajor", [0, 1])
@pytest.mark.parametrize("tile", [20, 21, 22, 23, 24, 25])
def test_dgemm_dl_matches_numpy(M, N, K, b_nmajor, tile):
    rng = np.random.default_rng(M * 5 + N * 11 + K)
    A = rng.normal(size=(M, K)); Bm = rng.normal(size=(K, N)); C0 = rng.normal(size=(M, N))
    Bdev = Bm.T.copy() if b_nmajor else Bm
    got = _gemm(A, Bdev, C0, -1.0, 1.0, b_nmajor, tile=tile)
    want = C0 - A @ Bm
    bound = 1e-12 * (np.abs(A) @ np.abs(Bm) + np.abs(C0))
    assert np.all(np.abs(got - want) <= bound)


@pytest.mark.parametrize("tile", [20, 21, 23, 24, 25])
def test_dgemm_dl_lower_only(tile):
    rng = np.random.default_rng(5)
    n, k = 900, 128
    A = rng.normal(size=(n, k)); C0 = rng.normal(size=(n, n))
    got = _gemm(A, A, C0, -1.0, 1.0, 1, lower_only=1, tile=tile)
    want = C0 - A @ A.T
    low = np.tril_indices(n)
    assert np.allclose(got[low], want[low], rtol=0, atol=1e-10)
    assert np.array_equal(got[0:64, 768:], C0[0:64, 768:])


def test_dgemm_dl_rejects_ragged_k():
    from glmmrmcml_amd import _lib
    with pytest.raises(_lib.McmlError):
        _gemm(np.ones((32, 17)), np.ones((17, 32)), np.zeros((32, 32)), 1.0, 0.0, 0, tile=20)


# ---- the banded zero-skipping kernel of the HMC products (dgemm_band.h) ----
def _band(A, B):
    from glmmrmcml_amd import _lib
    L = _lib.lib()
    M, K = A.shape
    N = B.shape[1]
    A = np.asfortranarray(A); B = np.asfortranarray(B); out = np.zeros((M, N), order="F")
    tiles = C.c_int()
    _lib.check(L.glmmr_mcml_dbg_dgemm_band(M, N, K, A.ctypes.data_as(dp), M, B.ctypes.data_as(dp), K,
                                           out.ctypes.data_as(dp), M, C.byref(tiles)))
    return out, tiles.value


def _check_band(A, B):
    got, tiles = _band(A, B)
    want = A @ B
    bound = 1e-12 * (np.abs(A) @ np.abs(B)) + 1e-300
    assert np.all(np.abs(got - want) <= bound)
    return tiles


# N = 1024 at M = 5000 runs the paired decomposition (one workgroup per CU, no partial sums); every other shape the
# streamed one (the (band, K tile) space cut into equal runs + k_band_reduce): N = 128 is one rank of the chain-sharded
# config 3 (1024 chains over 8 GPUs), N = 512 / 256 its 2- and 4-GPU forms
@pytest.mark.parametrize("M,N,K", [(5000, 1024, 5000), (80, 128, 32), (81, 129, 33), (333, 77, 250), (1000, 1, 999),
                                   (160, 300, 2000), (2000, 256, 2000), (5000, 128, 5000), (5000, 512, 5000),
                                   (5000, 256, 5000), (5000, 896, 5000)])
def test_band_lower_triangular(M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = np.tril(rng.normal(size=(M, K)))
    tiles = _check_band(A, rng.normal(size=(K, N)))
    nb, kt = (M + 79) // 80, (K + 31) // 32
    assert tiles <= nb * kt
    if M == K and M >= 1000:
        assert tiles < 0.6 * nb * kt                        # about half the K tiles are skipped


@pytest.mark.parametrize("M,N,K", [(5000, 1024, 5000), (333, 77, 250), (81, 129, 33)])
def test_band_upper_triangular_and_dense(M, N, K):
    rng = np.random.default_rng(7 * M + N)
    B = rng.normal(size=(K, N))
    _check_band(np.triu(rng.normal(size=(M, K))), B)
    tiles = _check_band(rng.normal(size=(M, K)), B)
    assert tiles == ((M + 79) // 80) * ((K + 31) // 32)      # nothing to skip


def test_band_general_band_and_empty_bands():
    rng = np.random.default_rng(99)
    M, K, N = 1300, 1700, 200
    A = rng.normal(size=(M, K))
    i, k = np.indices((M, K))
    A[np.abs(k - (i * K) // M) > 150] = 0.0                  # a diagonal band of half-width 150
    A[400:560, :] = 0.0                                      # two whole 80-row bands of zeros
    tiles = _check_band(A, rng.normal(size=(K, N)))
    assert tiles < 0.35 * ((M + 79) // 80) * ((K + 31) // 32)
    got, _ = _band(A, rng.normal(size=(K, N)))
    assert np.all(got[400:560, :] == 0.0)


def test_band_streamed_reduction_is_bit_reproducible():
    """the second stage sums a band's pieces in K order: two launches give identical bits"""
    rng = np.random.default_rng(5)
    A = np.tril(rng.normal(size=(2500, 2500)))
    B = rng.normal(size=(2500, 128))
    g1, _ = _band(A, B)
    g2, _ = _band(A, B)
    assert np.array_equal(g1, g2)
    g3, _ = _band(np.triu(A.T), B)
    g4, _ = _band(np.triu(A.T), B)
    assert np.array_equal(g3, g4)


@pytest.mark.gpu
def test_staged_host_device_copies_round_trip():
    """every host <-> device copy above 16 KB goes through the library's pinned staging buffer in chunks (common.hip):
    small (the runtime's own path), one chunk, many chunks with pitches on both sides, a single column longer than the
    buffer (piecewise), exactly at the chunk boundaries -- bytes in = bytes out, the host padding untouched"""
    from glmmrmcml_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(3)
    for rows, cols, pad_in, pad_out in ((7, 3, 2, 0), (2048, 1, 0, 0), (1000, 3000, 3, 5), (2 * 1024 * 1024, 1, 0, 0),
                                        (2 * 1024 * 1024 + 17, 2, 1, 0), (3 * 1024 * 1024 + 5, 1, 0, 0), (4096, 512, 0, 0)):
        a = np.asfortranarray(rng.normal(size=(rows + pad_in, cols)))
        out = np.full((rows + pad_out, cols), -7.0, order="F")
        _lib.check(L.glmmr_mcml_dbg_copy_roundtrip(a.ctypes.data_as(dp), C.c_longlong(rows + pad_in), out.ctypes.data_as(dp),
                                                   C.c_longlong(rows + pad_out), C.c_longlong(rows), C.c_longlong(cols)))
        assert np.array_equal(out[:rows], a[:rows]), (rows, cols)
        assert np.all(out[rows:] == -7.0), (rows, cols)
