"""GPU parity: HIP Gram build + normal-equation solve (through the C ABI) vs the float64 oracle.
Tolerance (BASELINE.json north_star): regression coefficients 1e-4, max|b-b_ref| / max|b_ref|,
b_ref = numpy.linalg.lstsq in float64 on the same float32 data (parity unpinned: see
oracle/wm_oracle.c -- tensorflow/keras, where the reference's fit lives, are absent)."""
import numpy as np
import pytest
import torch

from conftest import relinf
from cases import REG_CASES

pytestmark = pytest.mark.gpu
COEF_TOL = 1e-4


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("case", REG_CASES, ids=[c[0] for c in REG_CASES])
def test_gram_and_solve_match_oracle(case, regression_golden):
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    name, n, F, P, seed, dist = case
    X, Y, _ = synth.regression_problem(n, F, P, seed=seed, dist=dist)
    acc = ops.gram(_dev(X), _dev(Y))
    ref = ocpu.gram(X, Y)
    a = acc.cpu().numpy()
    assert a[F, F] == n
    assert np.array_equal(a, a.T)
    assert relinf(a, ref) < 2e-6          # float32 products, float64 combination
    beta = ops.normal_solve(acc, F, P).cpu().numpy()
    assert relinf(beta, regression_golden[f"{name}/beta_lstsq"]) < COEF_TOL
    assert relinf(beta, ocpu.normal_solve(ref, F, P)) < COEF_TOL
    # the solve kernel alone, fed the oracle's Gram, agrees to float64 accuracy
    b2 = ops.normal_solve(_dev(ref), F, P).cpu().numpy()
    assert relinf(b2, ocpu.normal_solve(ref, F, P)) < 1e-9


@pytest.mark.parametrize("n", [1, 3, 15, 16, 17, 63, 1000, 16 * 8 * 3 + 5, 70001])
def test_gram_f16_ragged_row_counts(n):
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    X, Y, _ = synth.regression_problem(n, 16, 1, seed=33)
    a = ops.gram(_dev(X), _dev(Y)).cpu().numpy()
    assert relinf(a, ocpu.gram(X, Y)) < 2e-6
    assert a[16, 16] == n


def test_gram_f16_asymmetric_operands_catch_layout_swaps():
    """Distinct integer features: exact in float32, so the Gram must be bit-exact; a swapped
    MFMA operand/result map would transpose or permute entries."""
    from weather_model_amd import ops
    from oracle import cpu as ocpu
    n = 257
    r = np.arange(n, dtype=np.float32)[:, None]
    f = np.arange(16, dtype=np.float32)[None, :]
    X = ((r * 7 + f * f * 3 + f) % 13) + f          # asymmetric in (row, feature)
    Y = ((r * 5) % 11) - 3.0
    a = ops.gram(_dev(X), _dev(Y)).cpu().numpy()
    assert np.array_equal(a, ocpu.gram(X, Y))


def test_gram_is_a_running_sum_and_reproducible():
    from weather_model_amd import ops, synth
    X, Y, _ = synth.regression_problem(50000, 16, 1, seed=5)
    Xd, Yd = _dev(X), _dev(Y)
    whole = ops.gram(Xd, Yd)
    acc = ops.gram(Xd[:20000], Yd[:20000])
    acc = ops.gram(Xd[20000:], Yd[20000:], acc)
    # float32 products summed in float32 for up to 1024 rows per accumulator (include/wm.h), then
    # float64: two splits of the rows associate those float32 sums differently
    assert relinf(acc.cpu().numpy(), whole.cpu().numpy()) < 1e-7
    assert torch.equal(ops.gram(Xd, Yd), whole)           # bitwise reproducible
    # rows may arrive in any order: the sufficient statistic is permutation invariant
    perm = torch.randperm(50000, device="cuda")
    assert relinf(ops.gram(Xd[perm], Yd[perm]).cpu().numpy(), whole.cpu().numpy()) < 1e-6


def test_gram_unaligned_and_generic_path_agree():
    from weather_model_amd import ops, synth
    X, Y, _ = synth.regression_problem(4099, 16, 1, seed=8)
    Xd, Yd = _dev(X), _dev(Y)
    fast = ops.gram(Xd, Yd)
    # a float32 view shifted by one element is 4-byte aligned only -> generic tile kernel
    buf = torch.empty(X.size + 1, device="cuda", dtype=torch.float32)
    buf[1:] = Xd.reshape(-1)
    slow = ops.gram(buf[1:].view(4099, 16), Yd)
    assert relinf(slow.cpu().numpy(), fast.cpu().numpy()) < 1e-6


def test_solve_options_and_singular_systems():
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    X, Y, _ = synth.regression_problem(3000, 5, 2, seed=3)
    acc = ops.gram(_dev(X), _dev(Y))
    ref = ocpu.gram(X, Y)
    for fi in (True, False):
        for ridge in (0.0, 0.5):
            b = ops.normal_solve(acc, 5, 2, fit_intercept=fi, ridge=ridge).cpu().numpy()
            assert relinf(b, ocpu.normal_solve(ref, 5, 2, fi, ridge)) < COEF_TOL
    Xc = X.copy()
    Xc[:, 4] = Xc[:, 3]                        # collinear -> not positive definite
    accc = ops.gram(_dev(Xc), _dev(Y))
    with pytest.raises(RuntimeError, match="positive definite"):
        ops.normal_solve(accc, 5, 2)
    assert np.isnan(ops.normal_solve(accc, 5, 2, check=False).cpu().numpy()).all()
    assert np.isfinite(ops.normal_solve(accc, 5, 2, ridge=1e-3).cpu().numpy()).all()
    # empty input is legal and leaves acc untouched
    z = ops.gram(torch.zeros(0, 5, device="cuda"), torch.zeros(0, 2, device="cuda"))
    assert float(z.abs().sum()) == 0.0


def test_device_generator_matches_host_generator():
    from weather_model_amd import ops, synth
    u = ops.synth_uniform((1000, 16), seed=9, lo=-1.0, hi=3.0, offset=77).cpu().numpy()
    assert np.array_equal(u, synth.uniform(9, (1000, 16), -1.0, 3.0, offset=77))
    beta = synth.normal(1001, (16, 1))
    Xd, Yd = ops.synth_regression(5000, 16, 1, seed=1, beta=torch.from_numpy(beta))
    Xh, Yh, _ = synth.regression_problem(5000, 16, 1, seed=1)
    assert np.array_equal(Xd.cpu().numpy(), Xh)
    assert np.abs(Yd.cpu().numpy() - Yh).max() < 1e-4       # Box-Muller: float32 vs float64 libm
    # paired draws (dist 2): the row kernel (F = 64, 16), the generic kernel (F = 6, P = 2), an odd
    # row offset -- bit for bit the host stream
    for F, P, off in ((64, 1, 0), (16, 1, 12345), (6, 2, 7)):
        beta = synth.normal(1003, (F, P))
        Xd, Yd = ops.synth_regression(3000, F, P, seed=3, beta=torch.from_numpy(beta), dist="uniform_paired",
                                      row_offset=off)
        Xh, Yh, _ = synth.regression_problem(3000, F, P, seed=3, dist="uniform_paired", row_offset=off)
        assert np.array_equal(Xd.cpu().numpy(), Xh)
        assert np.abs(Yd.cpu().numpy() - Yh).max() < 1e-4
    u = synth.uniform_paired01(5, 200_000)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 5e-3
    assert abs(np.corrcoef(u[0::2], u[1::2])[0, 1]) < 1e-2      # the two draws of a hash
    # one 32-bit hash per element (dist 3, config 5's generator): same three kernels, plus a row
    # offset whose element indices cross 2^32 (the high half of the index enters the key)
    for F, P, off in ((64, 1, 0), (16, 1, 12345), (6, 2, 7), (64, 1, (1 << 32) // 64 - 1500)):
        beta = synth.normal(1003, (F, P))
        Xd, Yd = ops.synth_regression(3000, F, P, seed=3, beta=torch.from_numpy(beta), dist="uniform32",
                                      row_offset=off)
        Xh, Yh, _ = synth.regression_problem(3000, F, P, seed=3, dist="uniform32", row_offset=off)
        assert np.array_equal(Xd.cpu().numpy(), Xh)
        assert np.abs(Yd.cpu().numpy() - Yh).max() < 1e-4
    u = synth.uniform32_01(5, 400_000)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 3e-3
    assert abs(u.var() - 1.0 / 12.0) < 1e-3
    for lag in (1, 2, 3, 4, 64):                                # neighbours in a row, and one row apart
        assert abs(np.corrcoef(u[:-lag], u[lag:])[0, 1]) < 1e-2
    assert abs(np.corrcoef(u, synth.uniform32_01(6, 400_000))[0, 1]) < 1e-2      # another seed


def test_full_size_property_recovers_planted_coefficients():
    """Size-independent property at a large row count (2e7 rows generated on the device):
    the fit recovers the planted beta within the statistical error of the noise, and the
    Gram of a row-doubled input is exactly twice the Gram (linearity)."""
    from weather_model_amd import ops, synth
    n = 20_000_000
    beta_true = synth.normal(1001, (16, 1))
    X, Y = ops.synth_regression(n, 16, 1, seed=1, beta=torch.from_numpy(beta_true))
    acc = ops.gram(X, Y)
    b = ops.normal_solve(acc, 16, 1).cpu().numpy()
    assert np.abs(b[:16] - beta_true).max() < 1e-4
    assert abs(b[16, 0] - 0.5) < 1e-4
    acc2 = ops.gram(X, Y, acc.clone())
    assert relinf(acc2.cpu().numpy(), 2.0 * acc.cpu().numpy()) < 1e-12
    # agreement with the float64 oracle on a slice of the same device data
    from oracle import cpu as ocpu
    Xs, Ys = X[:200000].cpu().numpy(), Y[:200000].cpu().numpy()
    assert relinf(ops.gram(X[:200000], Y[:200000]).cpu().numpy(), ocpu.gram(Xs, Ys)) < 2e-6


@pytest.mark.parametrize("n", [1, 3, 4, 5, 31, 257, 4 * 8 * 64 * 3 + 7, 50001])
def test_gram_x64_ragged_row_counts(n):
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    X, Y, _ = synth.regression_problem(n, 64, 1, seed=41)
    a = ops.gram(_dev(X), _dev(Y)).cpu().numpy()
    assert relinf(a, ocpu.gram(X, Y)) < 2e-6
    assert a[64, 64] == n
    assert np.array_equal(a, a.T)


def test_gram_x64_asymmetric_integers_are_exact():
    """Small distinct integers: every product and sum is exact in float32, so the 66x66 Gram
    must equal the oracle bit for bit; any operand/result map slip permutes entries."""
    from weather_model_amd import ops
    from oracle import cpu as ocpu
    n = 1029
    r = np.arange(n, dtype=np.float32)[:, None]
    f = np.arange(64, dtype=np.float32)[None, :]
    X = ((r * 7 + f * f * 3 + f * 5) % 11) + (f % 7)
    Y = ((r * 5) % 13) - 4.0
    a = ops.gram(_dev(X), _dev(Y)).cpu().numpy()
    assert np.array_equal(a, ocpu.gram(X, Y))


# ---------------------------------------------------------------------------------------------
# round 3: the VALU stream for narrow problems (gram_small.hip) and the LDS-staged blocked SYRK
# for everything else (gram_syrk.hip) -- the reference's own shapes are F=2,P=1
# (multipleRegression.py:35,273-286) and F=432,P=216 (singleRegression.py:21,33,74-85)

SMALL_SHAPES = [(1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1),
                (4, 2), (6, 3), (8, 4), (10, 5), (5, 2), (5, 3)]


def _int_problem(n, F, P):
    """Small distinct integers, asymmetric in (row, column): products and sums are exact in
    float32, so any Gram kernel must reproduce the float64 oracle bit for bit."""
    r = np.arange(n, dtype=np.float32)[:, None]
    f = np.arange(F, dtype=np.float32)[None, :]
    p = np.arange(P, dtype=np.float32)[None, :]
    X = ((r * 7 + f * f * 3 + f * 5) % 11) + (f % 7)
    Y = ((r * 5 + p * 3) % 13) - 4.0 + (p % 5)
    return np.ascontiguousarray(X, np.float32), np.ascontiguousarray(Y, np.float32)


@pytest.mark.parametrize("F,P", SMALL_SHAPES, ids=[f"f{f}p{p}" for f, p in SMALL_SHAPES])
def test_gram_small_every_instantiated_shape(F, P):
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    # exact integers: layout slips (row/column order, lane-to-row map) permute entries
    for n in (1, 67, 64 * 4 * 8 * 2 + 259):
        X, Y = _int_problem(n, F, P)
        a = ops.gram(_dev(X), _dev(Y)).cpu().numpy()
        assert np.array_equal(a, ocpu.gram(X, Y)), (F, P, n)
    # random rows, enough of them for several batches per wave and a float64 flush
    n = 3_000_017
    X, Y, _ = synth.regression_problem(n, F, P, seed=100 + F * 16 + P)
    Xd, Yd = _dev(X), _dev(Y)
    a = ops.gram(Xd, Yd)
    ref = ocpu.gram(X, Y)
    assert relinf(a.cpu().numpy(), ref) < 2e-6
    assert torch.equal(ops.gram(Xd, Yd), a)                 # bitwise reproducible
    beta = ops.normal_solve(a, F, P).cpu().numpy()
    assert relinf(beta, ocpu.normal_solve(ref, F, P)) < COEF_TOL


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 12 * 1024 + 1])
def test_gram_small_ragged_rows_reference_shape(n):
    """F=2, P=1 (multipleRegression's own fit): every row count around the batch boundaries."""
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    X, Y, _ = synth.regression_problem(n, 2, 1, seed=77)
    a = ops.gram(_dev(X), _dev(Y)).cpu().numpy()
    assert a[2, 2] == n
    assert relinf(a, ocpu.gram(X, Y)) < 2e-6


def test_gram_small_unaligned_falls_back_and_agrees():
    from weather_model_amd import ops, synth
    X, Y, _ = synth.regression_problem(5001, 2, 1, seed=78)
    Xd, Yd = _dev(X), _dev(Y)
    fast = ops.gram(Xd, Yd)
    buf = torch.empty(X.size + 1, device="cuda", dtype=torch.float32)
    buf[1:] = Xd.reshape(-1)
    slow = ops.gram(buf[1:].view(5001, 2), Yd)       # 4-byte aligned X: generic kernel
    assert relinf(slow.cpu().numpy(), fast.cpu().numpy()) < 1e-6


SYRK_SHAPES = [
    # F, P: one tile (4-way k split); two tiles; three tiles in one superblock; two superblocks
    # with a ragged last block; scalar fill (P % 4 != 0, F % 4 != 0); the reference's shape
    (24, 8), (60, 4), (100, 12), (130, 2), (33, 7), (200, 56), (432, 216),
    # F % 4 == 0 with ANY P: X by LDS-direct loads, the P + 1 columns of Y and of the ones by hand
    # (P = 1, 2, 6; Y and the ones column in the second superblock; Y straddling two superblocks)
    (32, 1), (16, 2), (12, 6), (128, 3), (124, 9),
]


@pytest.mark.parametrize("F,P", SYRK_SHAPES, ids=[f"f{f}p{p}" for f, p in SYRK_SHAPES])
def test_gram_syrk_shapes_exact_and_random(F, P):
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    for n in (1, 63, 64, 65, 2 * 64 * 16 + 77):
        X, Y = _int_problem(n, F, P)
        a = ops.gram(_dev(X), _dev(Y)).cpu().numpy()
        ref = ocpu.gram(X, Y)
        assert np.array_equal(a, ref), (F, P, n, np.argwhere(a != ref)[:4])
    n = 40_000 if F < 400 else 9_000
    X, Y, _ = synth.regression_problem(n, F, P, seed=200 + F, dist="normal")
    Xd, Yd = _dev(X), _dev(Y)
    a = ops.gram(Xd, Yd)
    ref = ocpu.gram(X, Y)
    assert relinf(a.cpu().numpy(), ref) < 2e-6
    assert torch.equal(ops.gram(Xd, Yd), a)
    # running sum over two row chunks
    h = n // 3
    acc = ops.gram(Xd[:h], Yd[:h])
    acc = ops.gram(Xd[h:], Yd[h:], acc)
    assert relinf(acc.cpu().numpy(), ref) < 2e-6


# the blocked multi-workgroup solve (solve_blocked.hip): every system beyond the one-wave kernel
BLOCKED_SOLVES = [(25, 1), (40, 5), (64, 8), (65, 3), (100, 70), (128, 64), (200, 130), (432, 216)]


@pytest.mark.parametrize("F,P", BLOCKED_SOLVES, ids=[f"f{f}p{p}" for f, p in BLOCKED_SOLVES])
def test_blocked_solve_matches_float64_oracle(F, P):
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    n = 6 * F + 50
    X, Y, _ = synth.regression_problem(n, F, P, seed=300 + F, dist="normal")
    ref = ocpu.gram(X, Y)
    acc = _dev(ref)
    for fi in (True, False):
        for ridge in (0.0, 0.25):
            b = ops.normal_solve(acc, F, P, fit_intercept=fi, ridge=ridge).cpu().numpy()
            assert relinf(b, ocpu.normal_solve(ref, F, P, fi, ridge)) < 1e-9, (fi, ridge)
    # and against numpy's least squares on the same rows (the coefficient tolerance of north_star)
    A = np.concatenate([X.astype(np.float64), np.ones((n, 1))], 1)
    ls = np.linalg.lstsq(A, Y.astype(np.float64), rcond=None)[0]
    assert relinf(ops.normal_solve(acc, F, P).cpu().numpy(), ls) < 1e-8
    # a collinear column in the LAST block: the pivot fails there, everything comes back NaN
    Xc = X.copy()
    Xc[:, F - 1] = Xc[:, 0]
    accc = _dev(ocpu.gram(Xc, Y))
    with pytest.raises(RuntimeError, match="positive definite"):
        ops.normal_solve(accc, F, P)
    assert np.isnan(ops.normal_solve(accc, F, P, check=False).cpu().numpy()).all()
    assert np.isfinite(ops.normal_solve(accc, F, P, ridge=1e-3).cpu().numpy()).all()


@pytest.mark.parametrize("F,P", [(24, 8), (100, 12), (200, 56), (432, 216), (32, 1), (124, 9)],
                         ids=["one_pair", "diag_superblock", "six_pairs", "reference_shape", "p1_by_hand",
                              "y_straddles"])
def test_gram_syrk_balanced_schedule_large_n(F, P):
    """Row counts large enough for the balanced persistent schedule (mode 1 of gram_syrk.hip):
    exact on integers (any slip in the stage dealing or the slab map loses or doubles rows), and
    within float32 product accuracy of the float64 oracle on random rows."""
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    n = 140_000 + 37
    X, Y = _int_problem(n, F, P)
    Xd, Yd = _dev(X), _dev(Y)
    a = ops.gram(Xd, Yd)
    ref = ocpu.gram(X, Y)
    assert np.array_equal(a.cpu().numpy(), ref)
    assert torch.equal(ops.gram(Xd, Yd), a)
    X, Y, _ = synth.regression_problem(n, F, P, seed=400 + F, dist="uniform")
    assert relinf(ops.gram(_dev(X), _dev(Y)).cpu().numpy(), ocpu.gram(X, Y)) < 2e-6


@pytest.mark.parametrize("F,P", [(16, 8), (12, 6), (14, 7), (24, 12), (40, 20), (33, 7), (62, 1), (15, 1),
                                 (18, 9), (20, 10), (22, 11), (32, 16), (32, 1), (9, 1), (28, 3), (25, 5),
                                 (3, 2), (1, 3), (21, 12)], ids=lambda v: str(v))
def test_gram_mid_kernel_alignments_and_tails(F, P):
    """gram_mid.hip (15 < F + P, F + P + 1 <= 64): vector-load and dword-load variants, both column
    widths (D <= 32, D <= 64), pointers moved off their 16- / 8-byte alignment, row counts around the
    k-step (4 rows), the in-flight batch and the per-wave ranges; exact on integers, and against the
    float64 oracle on random rows.  Every shape with ceil(F/4) + ceil(P/4) <= 9 and P <= 12 that the
    narrow-row and the aligned quad kernel leave -- the family F = 2p, P = p for p = 6 .. 12, one target
    with 9 .. 32 predictors, any of them off its alignment -- now takes gram_quadu_kernel<RX, RY>
    (gram_quad.hip: padded pieces, dword-aligned vector loads that run into the next row, last row
    read element by element) -- same checks; the row counts below put the last row in a full tile,
    in a partial one and alone."""
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    for n in (1, 3, 4, 5, 15, 16, 17, 63, 64, 65, 8191, 8192 * 4 + 3, 2048 * 128, 300_007):
        X, Y = _int_problem(n, F, P)
        ref = ocpu.gram(X, Y)
        a = ops.gram(_dev(X), _dev(Y)).cpu().numpy()
        assert np.array_equal(a, ref), (F, P, n, np.argwhere(a != ref)[:4])
        for off in (1, 2):                        # 4- and 8-byte offsets from the allocation
            bx = torch.zeros(n * F + off, device="cuda")
            by = torch.zeros(n * P + off, device="cuda")
            bx[off:] = _dev(X).reshape(-1)
            by[off:] = _dev(Y).reshape(-1)
            b = ops.gram(bx[off:].view(n, F), by[off:].view(n, P)).cpu().numpy()
            assert np.array_equal(b, ref), (F, P, n, off)
    n = 200_000
    X, Y, _ = synth.regression_problem(n, F, P, seed=700 + F, dist="normal")
    Xd, Yd = _dev(X), _dev(Y)
    a = ops.gram(Xd, Yd)
    assert relinf(a.cpu().numpy(), ocpu.gram(X, Y)) < 2e-6
    assert torch.equal(ops.gram(Xd, Yd), a)                       # bitwise reproducible


@pytest.mark.parametrize("shape", [
    (2, 1, 100_003, True), (16, 8, 70_001, True), (12, 6, 33_333, True), (32, 1, 50_000, False),
    (16, 8, 40_000, False), (3, 2, 999, True), (64, 32, 20_011, True), (5, 20, 9_000, True),
    (24, 12, 25_037, True), (8, 4, 64 * 300, True), (40, 16, 12_345, True), (16, 8, 63, True),
    (256, 2, 5_001, True), (4, 4, 77_777, True),
    (432, 216, 3_000, True),          # beyond the fused kernel: library GEMM + wm_residual_sums_f32
])
def test_linear_metrics_match_float64_restatement(shape):
    """loss / mae / msle of the current coefficients (multipleRegression.py:292-296) in one fused pass
    against a float64 numpy restatement; float32 prediction rounding bounds the difference."""
    from weather_model_amd import ops
    F, P, n, msle = shape
    rng = np.random.default_rng(F * 100 + P)
    X = rng.uniform(0.0, 1.0, (n, F)).astype(np.float32)
    beta = rng.uniform(-0.5, 0.5, (F + 1, P))
    Y = (X.astype(np.float64) @ beta[:F] + beta[F] + rng.normal(0, 0.05, (n, P))).astype(np.float32)
    Xd = torch.from_numpy(X).cuda()
    sums = ops.linear_metrics(Xd, torch.from_numpy(Y).cuda(), torch.from_numpy(beta).cuda(),
                              want_msle=msle)
    pred = (X.astype(np.float64) @ beta[:F] + beta[F]).astype(np.float32).astype(np.float64)
    d = pred - Y
    eps = np.float32(1e-7)
    e = np.log1p(np.maximum(pred, eps)) - np.log1p(np.maximum(Y.astype(np.float64), eps))
    want = np.array([(d * d).sum(), np.abs(d).sum(), (e * e).sum() if msle else 0.0, n * P])
    got = sums.cpu().numpy()
    tol = 2e-4 if F > 64 else 2e-6        # the GEMM path rounds beta and the products to float32
    assert got[3] == want[3]
    np.testing.assert_allclose(got[:3], want[:3], rtol=tol, atol=1e-12)
    # accumulating: a second call over a sub-range adds to the same sums, misaligned X included
    again = ops.linear_metrics(Xd[1:n // 2], torch.from_numpy(Y[1:n // 2]).cuda(),
                               torch.from_numpy(beta).cuda(), sums=sums.clone(), want_msle=msle)
    d2 = d[1:n // 2]
    np.testing.assert_allclose(float(again[0]), want[0] + (d2 * d2).sum(), rtol=tol)
    assert float(again[3]) == want[3] + d2.size
    # bitwise reproducible
    b = ops.linear_metrics(Xd, torch.from_numpy(Y).cuda(), torch.from_numpy(beta).cuda(), want_msle=msle)
    assert torch.equal(b, sums)


@pytest.mark.parametrize("F,P", [(1, 1), (2, 1), (4, 2), (6, 3), (8, 4), (10, 5), (5, 3), (8, 1), (3, 1), (7, 1),
                                 (16, 8), (12, 6), (32, 16), (16, 4), (8, 8),
                                 (14, 7), (18, 9), (22, 11), (20, 10), (24, 12)], ids=lambda v: str(v))
def test_linear_metrics_stream_kernels_tails_and_far_branch(F, P):
    """linmetrics.hip's narrow-row kernel (F + P <= 15, gram_small's memory side) and the prefetching
    tile kernel: row counts around a lane's batch (64 R U rows), a wave's tile (64 rows) and more than
    one batch per wave; targets UNRELATED to the predictions, so that many outputs take the far form
    of log1p(a) - log1p(b) (the two differ by more than a factor 1.5) beside near ones in the same
    wave, plus values below eps and beyond the 1e37 clamp; msle on and off; pointers off their
    alignment take another kernel and must agree to rounding.  F=18,P=9 .. F=24,P=12 take the
    float64-MFMA kernel (flat LDS tiles, trimmed last feature block, partial last tile with a row
    tail that is not a whole 16-byte piece)."""
    from weather_model_amd import ops
    rng = np.random.default_rng(F * 31 + P)
    beta = rng.uniform(-0.5, 0.5, (F + 1, P))
    beta[F] += 1.0
    bd = torch.from_numpy(beta).cuda()
    eps = np.float32(1e-7)
    for n in (1, 63, 64, 65, 511, 512, 513, 1023, 4097, 2048 * 64 * 4 + 77, 700_001):
        X = rng.uniform(0.0, 1.0, (n, F)).astype(np.float32)
        pred = (X.astype(np.float64) @ beta[:F] + beta[F]).astype(np.float32).astype(np.float64)
        Y = pred.astype(np.float32)
        far = rng.uniform(size=Y.shape) < 0.3
        Y[far] = rng.uniform(-0.5, 6.0, int(far.sum())).astype(np.float32)      # far form, some below eps
        Y += rng.normal(0, 0.01, Y.shape).astype(np.float32)
        if n > 100:
            Y[7, 0] = np.float32(3e38)                                          # beyond the clamp
        Yd = Y.astype(np.float64)
        d = pred - Yd
        e = np.log1p(np.clip(pred, eps, 1e37)) - np.log1p(np.clip(Yd, eps, np.float64(np.float32(1e37))))
        for msle in (True, False):
            want = np.array([(d * d).sum(), np.abs(d).sum(), (e * e).sum() if msle else 0.0, n * P])
            Xd, Yt = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
            got = ops.linear_metrics(Xd, Yt, bd, want_msle=msle).cpu().numpy()
            assert got[3] == want[3]
            np.testing.assert_allclose(got[:3], want[:3], rtol=3e-6, atol=1e-12, err_msg=f"{F},{P},{n},{msle}")
            assert torch.equal(ops.linear_metrics(Xd, Yt, bd, want_msle=msle).cpu(), torch.from_numpy(got))
            if n in (513, 4097):
                bx = torch.zeros(n * F + 1, device="cuda")
                by = torch.zeros(n * P + 1, device="cuda")
                bx[1:] = Xd.reshape(-1)
                by[1:] = Yt.reshape(-1)
                off = ops.linear_metrics(bx[1:].view(n, F), by[1:].view(n, P), bd, want_msle=msle).cpu().numpy()
                np.testing.assert_allclose(off, got, rtol=3e-6, atol=1e-12)


@pytest.mark.parametrize("F,P", [(432, 216), (20, 8), (64, 64), (100, 128), (36, 220), (16, 4), (48, 36)],
                         ids=lambda v: str(v))
def test_linear_wide_mfma_metrics_and_predict(F, P):
    """wm_linear_metrics_wide_f32 (the reference's singleRegression map 432 -> 216,
    singleRegression.py:74-85, and every panel width 4 / 8 / 14 tiles): predictions from the fp32
    MFMA accumulators against a float64 product (float32 products and sums: 2e-5 of the largest
    prediction), residual sums against a float64 restatement on those float32 predictions' float64
    twins (2e-4: the predictions differ in their last bits), for row counts around the 32-row wave
    tile and the 128-row workgroup pass and for F that is not a multiple of the 16-index stage;
    predictions written and scored in one call equal the two separate calls bit for bit; targets far
    from the predictions exercise the repeat pass."""
    from weather_model_amd import ops
    rng = np.random.default_rng(F + P)
    beta = rng.uniform(-0.5, 0.5, (F + 1, P)) / np.sqrt(F)
    beta[F] = rng.uniform(0.5, 1.5, P)
    bd = torch.from_numpy(beta).cuda()
    eps = np.float32(1e-7)
    for n in (1, 31, 32, 33, 127, 128, 129, 1000, 20_011):
        X = rng.uniform(0.0, 1.0, (n, F)).astype(np.float32)
        ref = X.astype(np.float64) @ beta[:F] + beta[F]
        Y = (ref + rng.normal(0, 0.05, (n, P))).astype(np.float32)
        if n in (129, 1000):
            # targets far from the predictions (and below eps): the wave raises the flag and the
            # repeat pass replaces the msle sum with the one that has both forms
            far = rng.uniform(size=Y.shape) < (0.2 if n == 129 else 0.001)
            Y[far] = rng.uniform(-0.5, 9.0, int(far.sum())).astype(np.float32)
        Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
        _, pred = ops.linear_wide(Xd, bd, predict=True)
        pn = pred.cpu().numpy().astype(np.float64)
        assert np.abs(pn - ref).max() <= 2e-5 * np.abs(ref).max(), (F, P, n)
        d = ref - Y
        e = np.log1p(np.maximum(ref, eps)) - np.log1p(np.maximum(Y.astype(np.float64), eps))
        for msle in (True, False):
            want = np.array([(d * d).sum(), np.abs(d).sum(), (e * e).sum() if msle else 0.0, n * P])
            sums = torch.zeros(4, device="cuda", dtype=torch.float64)
            ops.linear_wide(Xd, bd, Yd, sums, want_msle=msle)
            got = sums.cpu().numpy()
            assert got[3] == want[3]
            np.testing.assert_allclose(got[:3], want[:3], rtol=2e-4, atol=1e-12, err_msg=f"{F},{P},{n},{msle}")
            both = torch.zeros(4, device="cuda", dtype=torch.float64)
            _, p2 = ops.linear_wide(Xd, bd, Yd, both, want_msle=msle, predict=True)
            assert torch.equal(p2, pred) and torch.equal(both, sums)
    # what the drop-ins call: ops.linear_metrics / ops.linear_predict route wide maps here
    if F * P >= 4096:
        assert torch.equal(ops.linear_predict(Xd, bd), pred)
    if not _lib_supported(F, P):
        assert torch.equal(ops.linear_metrics(Xd, Yd, bd, want_msle=False), sums)


def _lib_supported(F, P):
    from weather_model_amd import _lib
    return bool(_lib.lib().wm_linear_metrics_supported(F, P))


def test_linear_metrics_rejects_bad_shapes():
    from weather_model_amd import ops
    X = torch.zeros((10, 4), device="cuda")
    with pytest.raises(ValueError):
        ops.linear_metrics(X, torch.zeros((10, 3), device="cuda"),
                           torch.zeros((5, 2), device="cuda", dtype=torch.float64))
    with pytest.raises(ValueError):
        ops.linear_metrics(X, torch.zeros((10, 2), device="cuda"),
                           torch.zeros((5, 2), device="cuda", dtype=torch.float64),
                           sums=torch.zeros(3, device="cuda", dtype=torch.float64))
    s = ops.linear_metrics(X[:0], torch.zeros((0, 2), device="cuda"),
                           torch.zeros((5, 2), device="cuda", dtype=torch.float64))
    assert s.tolist() == [0.0, 0.0, 0.0, 0.0]


def test_linear_metrics_full_size_properties():
    """4e7 rows of F=16, P=8 (the size the Gram tests use): with Y = X beta + c + uniform noise of
    standard deviation s the fused pass must report mse = s^2 and mae = s sqrt(3) / 2 (the mean
    absolute value of that uniform) to sampling accuracy, the exact count, and the same sums whether
    the rows go through in one call or accumulate over uneven chunks (1e-12: different partial
    orders)."""
    from weather_model_amd import ops
    n, F, P, s = 40_000_000, 16, 8, 0.01
    X = ops.synth_uniform((n, F), seed=41)
    beta = torch.from_numpy(np.random.default_rng(7).uniform(-0.5, 0.5, (F + 1, P))).cuda()
    Y = ops.synth_uniform((n, P), seed=42, lo=-s * 3 ** 0.5, hi=s * 3 ** 0.5)
    step = 4_000_000
    for a in range(0, n, step):
        Y[a:a + step] += (X[a:a + step].double() @ beta[:F] + beta[F]).float()
    one = ops.linear_metrics(X, Y, beta).cpu().numpy()
    assert one[3] == n * P
    assert one[0] / one[3] == pytest.approx(s * s, rel=5e-3)
    assert one[1] / one[3] == pytest.approx(s * 3 ** 0.5 / 2, rel=5e-3)
    assert 0 < one[2] / one[3] < s * s            # log1p compresses: msle below mse for values >= 0
    acc = None
    for a, b in ((0, 13_000_001), (13_000_001, 13_000_064), (13_000_064, n)):
        acc = ops.linear_metrics(X[a:b], Y[a:b], beta, sums=acc)
    np.testing.assert_allclose(acc.cpu().numpy(), one, rtol=1e-12)


@pytest.mark.parametrize("F,P", [(12, 6), (20, 10), (22, 11), (10, 5), (6, 3)], ids=lambda v: str(v))
def test_gram_family_kernels_flush_at_full_size(F, P):
    """gram_quadu_kernel (padded pieces: F=12,P=6 with batches of 8 tiles, 20,10 of 4, 22,11 of 2) and
    gram_small_kernel's dense-load path (10,5 / 6,3) at 4e7 rows: more than 1024 tiles (rows) per wave,
    so the float32 sums are folded into float64 inside the loop; checked against a float64 product
    formed in chunks on the device, bitwise reproducible, exact count, symmetric.  The same rows give
    the solved coefficients' loss through the fused metrics pass: it must equal the closed form
    y'y - 2 b'X'y + b'X'Xb of the Gram to the float32 rounding of the predictions."""
    from weather_model_amd import ops
    n = 40_000_000
    Xd = ops.synth_uniform((n, F), seed=61)
    Yd = ops.synth_uniform((n, P), seed=62, lo=-0.02, hi=0.02)
    bt = torch.from_numpy(np.random.default_rng(F).uniform(-0.5, 0.5, (F + 1, P))).cuda()
    bt[F] += 1.0
    step = 2_000_000
    for s0 in range(0, n, step):
        Yd[s0:s0 + step] += (Xd[s0:s0 + step].double() @ bt[:F] + bt[F]).float()
    a = ops.gram(Xd, Yd)
    assert torch.equal(ops.gram(Xd, Yd), a)                       # bitwise reproducible
    D = F + 1 + P
    ref = torch.zeros((D, D), device="cuda", dtype=torch.float64)
    for s0 in range(0, n, step):
        Z = torch.cat([Xd[s0:s0 + step].double(),
                       torch.ones((min(step, n - s0), 1), device="cuda", dtype=torch.float64),
                       Yd[s0:s0 + step].double()], 1)
        ref += Z.T @ Z
    assert a[F, F].item() == n
    assert relinf(a.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert torch.equal(a, a.T)
    beta = ops.normal_solve(a, F, P)
    sums = ops.linear_metrics(Xd, Yd, beta).cpu().numpy()
    assert sums[3] == n * P
    r = ref.cpu().numpy()
    b = beta.cpu().numpy()
    A = r[:F + 1, :F + 1]
    B = r[:F + 1, F + 1:]
    C = r[F + 1:, F + 1:]
    sse = np.trace(C) - 2.0 * (b * B).sum() + (b * (A @ b)).sum()
    assert sums[0] == pytest.approx(sse, rel=2e-4)


@pytest.mark.parametrize("F,P", [(16, 8), (16, 4), (8, 8)], ids=lambda v: str(v))
def test_gram_quad_kernel_exact_tails_and_flush(F, P):
    """gram_quad.hip (rows of four equal pieces; eight variables at once = F=16, P=8): exact on
    integers for row counts around the 16-row tile, the 8-tile batch and more than one batch per
    wave (2048 waves x 128 rows); pointers off their alignment take another kernel and must agree;
    4e7 random rows cross the float32 -> float64 flush (1024 tiles per wave) and are checked against
    a float64 product formed in chunks on the device."""
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    for n in (1, 15, 16, 17, 127, 128, 129, 2048 * 128 + 5, 1_000_003):
        X, Y = _int_problem(n, F, P)
        ref = ocpu.gram(X, Y)
        a = ops.gram(_dev(X), _dev(Y)).cpu().numpy()
        assert np.array_equal(a, ref), (F, P, n, np.argwhere(a != ref)[:4])
    bx = torch.zeros(n * F + 1, device="cuda")
    by = torch.zeros(n * P + 1, device="cuda")
    bx[1:] = _dev(X).reshape(-1)
    by[1:] = _dev(Y).reshape(-1)
    assert np.array_equal(ops.gram(bx[1:].view(n, F), by[1:].view(n, P)).cpu().numpy(), ref)
    n = 40_000_000
    Xd = ops.synth_uniform((n, F), seed=51)
    Yd = ops.synth_uniform((n, P), seed=52, lo=-1.0, hi=2.0)
    a = ops.gram(Xd, Yd)
    assert torch.equal(ops.gram(Xd, Yd), a)                       # bitwise reproducible
    D = F + 1 + P
    ref = torch.zeros((D, D), device="cuda", dtype=torch.float64)
    for s0 in range(0, n, 2_000_000):
        Z = torch.cat([Xd[s0:s0 + 2_000_000].double(),
                       torch.ones((min(2_000_000, n - s0), 1), device="cuda", dtype=torch.float64),
                       Yd[s0:s0 + 2_000_000].double()], 1)
        ref += Z.T @ Z
    assert a[F, F].item() == n
    assert relinf(a.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert torch.equal(a, a.T)
