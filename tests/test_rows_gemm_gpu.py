"""msr3d_rows_gemm_split (csrc/rows_gemm_split.hip): the tall token GEMMs of an unfrozen backbone's SharedMLP
layers (/root/reference/model/pointnet2/pytorch_utils.py:9-60) on the bf16 pipe at fp32 accuracy, against
float64 -- tolerance 2e-6 of the product's scale (an fp32 GEMM's own rounding), both operand orientations,
ragged row counts, widths that are not a multiple of 16, zero-padded reduction widths."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(M, N, K, b_trans, lda=None, ldc=None, seed=0, with_stats=False):
    from msr3d_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(seed + M + 7 * N + 13 * K)
    lda = lda or K
    ldc = ldc or ((N + 3) // 4 * 4)
    A = torch.randn(M, lda, device="cuda", generator=g)
    B = torch.randn((K, N) if b_trans else (N, K), device="cuda", generator=g) * 0.3
    C = torch.full((M, ldc), float("nan"), device="cuda")
    st = _lib.current_stream_ptr(torch.device("cuda"))
    stats = torch.full(((M + 255) // 256, 2, N), float("nan"), device="cuda") if with_stats else None
    rc = _lib.load().msr3d_rows_gemm_split(M, N, K, ctypes.c_void_p(A.data_ptr()), lda, ctypes.c_void_p(B.data_ptr()),
                                           B.shape[1], int(b_trans), ctypes.c_void_p(C.data_ptr()), ldc,
                                           ctypes.c_void_p(stats.data_ptr() if with_stats else 0), None, st)
    assert rc == 0
    if with_stats:
        # per 256-row block: column sums and sums of squares of the STORED values (fp32 summation noise only)
        Cd = C[:, :N].double()
        pad = (-M) % 256
        Cp = torch.cat([Cd, torch.zeros(pad, N, device="cuda", dtype=torch.float64)]).view(-1, 256, N)
        s1, s2 = Cp.sum(1), (Cp * Cp).sum(1)
        assert float((stats[:, 0].double() - s1).abs().max() / Cp.abs().sum(1).max()) < 1e-6
        assert float((stats[:, 1].double() - s2).abs().max() / s2.max()) < 1e-6
    want = A[:, :K].double() @ (B.double() if b_trans else B.double().t())
    scale = (A[:, :K].double().abs() @ (B.double().abs() if b_trans else B.double().abs().t()))
    err = float(((C[:, :N].double() - want).abs() / scale).max())
    assert err < 2e-6, err
    if ldc > N:
        assert bool(torch.isnan(C[:, N:]).all())          # columns past N are not touched
    return err


@pytest.mark.parametrize("M,N,K,b_trans", [
    (40000, 64, 4, 0), (40000, 64, 64, 0), (40000, 128, 64, 0),          # sa1 forward
    (30000, 128, 132, 0), (30000, 128, 128, 0), (30000, 256, 128, 0),    # sa2 forward
    (40000, 64, 128, 1), (40000, 64, 64, 1),                             # sa1 d t = d z W
    (30000, 128, 128, 1), (30000, 132, 128, 1),                          # sa2
    (257, 144, 160, 0), (31, 20, 36, 1), (1, 4, 4, 0), (513, 200, 96, 0),
    (15360, 256, 264, 0), (15360, 512, 256, 0), (15360, 768, 512, 0),   # sa3 forward: super-slabs of 128
    (15360, 512, 768, 1), (15360, 256, 512, 1), (9000, 260, 256, 1), (300, 1000, 1000, 0),
    (70000, 128, 256, 1), (66000, 132, 192, 0)])                         # tall, 161..256-wide reduction
def test_rows_gemm_split_vs_float64(M, N, K, b_trans):
    _run(M, N, K, b_trans)


@pytest.mark.parametrize("M,N,K", [(40000, 64, 64), (30000, 128, 132), (30001, 256, 128), (257, 144, 160), (100, 20, 36),
                                   (15360, 512, 256), (1000, 768, 512)])
def test_rows_gemm_split_column_statistics(M, N, K):
    _run(M, N, K, 0, with_stats=True)


def test_rows_gemm_split_padded_pitches_and_reproducible():
    from msr3d_amd import _lib
    _run(9000, 64, 64, 0, lda=72, ldc=80)
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.randn(70000, 128, device="cuda", generator=g)
    B = torch.randn(128, 128, device="cuda", generator=g)
    st = _lib.current_stream_ptr(torch.device("cuda"))
    outs = []
    for _ in range(3):
        C = torch.empty(70000, 128, device="cuda")
        assert _lib.load().msr3d_rows_gemm_split(70000, 128, 128, ctypes.c_void_p(A.data_ptr()), 128,
                                                 ctypes.c_void_p(B.data_ptr()), 128, 0,
                                                 ctypes.c_void_p(C.data_ptr()), 128, None, None, st) == 0
        outs.append(C)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_rows_gemm_split_rejects_what_it_does_not_take():
    from msr3d_amd import _lib
    A = torch.zeros(64, 256, device="cuda")
    C = torch.zeros(64, 512, device="cuda")
    st = _lib.current_stream_ptr(torch.device("cuda"))
    f = _lib.load().msr3d_rows_gemm_split
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    assert f(64, 64, 1028, p(A), 1028, p(A), 1028, 0, p(C), 512, None, None, st) == -22   # K > 1024
    assert f(64, 1028, 64, p(A), 256, p(A), 256, 0, p(C), 1028, None, None, st) == -22    # N > 1024
    assert f(64, 64, 62, p(A), 256, p(A), 256, 0, p(C), 512, None, None, st) == -22       # K % 4


def test_rows_gemm_split_throughput():
    """The sa2 middle layer at 16 scenes x 60 objects: 491,520 rows x 128 -> 128; report time and HBM rate."""
    from msr3d_amd import _lib
    M, N, K = 491520, 128, 128
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    C = torch.empty(M, N, device="cuda")
    S = torch.empty((M + 255) // 256, 2, N, device="cuda")        # (with the statistics epilogue, as the layer runs it)
    st = _lib.current_stream_ptr(torch.device("cuda"))
    f = _lib.load().msr3d_rows_gemm_split
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    for _ in range(5):
        f(M, N, K, p(A), K, p(B), K, 0, p(C), N, p(S), None, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f(M, N, K, p(A), K, p(B), K, 0, p(C), N, p(S), None, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"\nrows_gemm_split {M} x {K} -> {N}: {us:.0f} us = {(M * (K + N) * 4) / us / 1e6:.2f} TB/s of rows, "
          f"{2 * M * N * K / us / 1e6:.0f} TFLOP/s fp32-equivalent")
    assert us < 400


def _wgrad(M, N, K, accumulate=False, ldy=None, ldx=None, seed=0):
    from msr3d_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(seed + M + 3 * N + 11 * K)
    ldy, ldx = ldy or N, ldx or K
    dy = torch.randn(M, ldy, device="cuda", generator=g)
    x = torch.randn(M, ldx, device="cuda", generator=g)
    dW0 = torch.randn(N, K, device="cuda", generator=g)
    dW = dW0.clone()
    ws = torch.empty(256 * N * K, device="cuda")
    st = _lib.current_stream_ptr(torch.device("cuda"))
    rc = _lib.load().msr3d_wgrad_rows_split(M, N, K, ctypes.c_void_p(dy.data_ptr()), ldy, ctypes.c_void_p(x.data_ptr()), ldx,
                                            ctypes.c_void_p(dW.data_ptr()), K, int(accumulate),
                                            ctypes.c_void_p(ws.data_ptr()), ws.numel(), None, st)
    assert rc == 0
    want = dy[:, :N].double().t() @ x[:, :K].double() + (dW0.double() if accumulate else 0)
    scale = dy[:, :N].double().abs().t() @ x[:, :K].double().abs() + 1.0
    err = float(((dW.double() - want).abs() / scale).max())
    assert err < 5e-6, err
    return dW


@pytest.mark.parametrize("M,N,K", [(983040, 64, 64), (200000, 64, 4), (200000, 128, 64), (150000, 128, 132),
                                   (150000, 256, 128), (70001, 128, 128), (700, 64, 64), (40, 20, 12)])
def test_wgrad_rows_split_vs_float64(M, N, K):
    _wgrad(M, N, K)


def test_wgrad_rows_split_accumulates_pitches_and_is_reproducible():
    _wgrad(50000, 64, 64, accumulate=True, ldy=72, ldx=68)
    a = _wgrad(300000, 128, 128, seed=3)
    b = _wgrad(300000, 128, 128, seed=3)
    assert torch.equal(a, b)


def test_wgrad_rows_split_throughput():
    from msr3d_amd import _lib
    M, N, K = 491520, 128, 128
    dy = torch.randn(M, N, device="cuda")
    x = torch.randn(M, K, device="cuda")
    dW = torch.empty(N, K, device="cuda")
    ws = torch.empty(256 * N * K, device="cuda")
    st = _lib.current_stream_ptr(torch.device("cuda"))
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    f = lambda: _lib.load().msr3d_wgrad_rows_split(M, N, K, p(dy), N, p(x), K, p(dW), K, 0, p(ws), ws.numel(), None, st)   # noqa: E731
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"\nwgrad_rows_split {M} rows, {N} x {K}: {us:.0f} us = {(M * (K + N) * 4) / us / 1e6:.2f} TB/s of rows")
    assert us < 600


@pytest.mark.parametrize("M,N,K", [(40000, 64, 64), (30000, 128, 128), (30001, 256, 128), (9000, 128, 64), (300, 64, 132)])
def test_operand_normalisation_on_load(M, N, K):
    """a_bn / x_bn: the products of relu(batch_norm(A)) without the normalised activation in memory -- against the
    same products of the explicitly normalised operand (fp32 formula of bn_train.hip) in float64."""
    from msr3d_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g) * 2 + 0.3
    B = torch.randn(N, K, device="cuda", generator=g) * 0.3
    pro = torch.stack([1 + 0.2 * torch.randn(K, device="cuda", generator=g), 0.3 * torch.randn(K, device="cuda", generator=g),
                       0.3 + 0.1 * torch.randn(K, device="cuda", generator=g),
                       0.5 + 0.1 * torch.rand(K, device="cuda", generator=g)]).contiguous()
    Y = torch.relu(torch.addcmul(pro[1], pro[0], (A - pro[2]) * pro[3]))          # the kernels' fp32 statement
    C = torch.empty(M, N, device="cuda")
    st = _lib.current_stream_ptr(torch.device("cuda"))
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    assert _lib.load().msr3d_rows_gemm_split(M, N, K, p(A), K, p(B), K, 0, p(C), N, None, p(pro), st) == 0
    want = Y.double() @ B.double().t()
    scale = Y.double().abs() @ B.double().abs().t() + 1e-3
    assert float(((C.double() - want).abs() / scale).max()) < 4e-6
    dy = torch.randn(M, N, device="cuda", generator=g)
    dW = torch.empty(N, K, device="cuda")
    ws = torch.empty(256 * N * K, device="cuda")
    assert _lib.load().msr3d_wgrad_rows_split(M, N, K, p(dy), N, p(A), K, p(dW), K, 0, p(ws), ws.numel(), p(pro), st) == 0
    want = dy.double().t() @ Y.double()
    scale = dy.double().abs().t() @ Y.double().abs() + 1.0
    assert float(((dW.double() - want).abs() / scale).max()) < 5e-6


@pytest.mark.parametrize("b_trans", [0, 1])
def test_weight_narrower_than_the_padded_operand(b_trans):
    """The first SharedMLP layer: rows zero-padded from 131 to 144 columns against the (128, 131) weight as it lies
    (ldb = 131 < K = 144), and its transpose product d t = d z W with 144 output columns of which 131 exist."""
    from msr3d_amd import _lib
    M, N, Kr, KP = 20000, 128, 131, 144
    g = torch.Generator(device="cuda").manual_seed(7 + b_trans)
    W = torch.randn(N, Kr, device="cuda", generator=g) * 0.3
    st = _lib.current_stream_ptr(torch.device("cuda"))
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    if not b_trans:
        A = torch.zeros(M, KP, device="cuda")
        A[:, :Kr] = torch.randn(M, Kr, device="cuda", generator=g)
        C = torch.empty(M, N, device="cuda")
        assert _lib.load().msr3d_rows_gemm_split(M, N, KP, p(A), KP, p(W), Kr, 0, p(C), N, None, None, st) == 0
        want = A[:, :Kr].double() @ W.double().t()
        scale = A[:, :Kr].double().abs() @ W.double().abs().t()
    else:
        A = torch.randn(M, N, device="cuda", generator=g)
        C = torch.full((M, KP), float("nan"), device="cuda")
        assert _lib.load().msr3d_rows_gemm_split(M, KP, N, p(A), N, p(W), Kr, 1, p(C), KP, None, None, st) == 0
        assert float(C[:, Kr:].abs().max()) == 0.0          # the padding columns of d t: exact zeros
        C = C[:, :Kr]
        want = A.double() @ W.double()
        scale = A.double().abs() @ W.double().abs()
    assert float(((C.double() - want).abs() / scale).max()) < 2e-6


def test_full_size_checksums_of_the_first_level():
    """At the bench's full row count (16 scenes x 60 objects x 32 centres x 32 neighbours = 983,040 rows), through
    size-independent properties: the column sums of C -- the kernel's own col_stats -- equal (column sums of A) B^T
    (a checksum of checksums), and the weight gradient over all rows equals the sum of the gradients over the two
    halves of the rows (linearity)."""
    from msr3d_amd import _lib
    M, N, K = 983040, 128, 64
    g = torch.Generator(device="cuda").manual_seed(11)
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(N, K, device="cuda", generator=g) * 0.3
    C = torch.empty(M, N, device="cuda")
    S = torch.empty((M + 255) // 256, 2, N, device="cuda")
    st = _lib.current_stream_ptr(torch.device("cuda"))
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    assert _lib.load().msr3d_rows_gemm_split(M, N, K, p(A), K, p(B), K, 0, p(C), N, p(S), None, st) == 0
    got = S[:, 0].double().sum(0)
    want = A.double().sum(0) @ B.double().t()
    scale = A.double().abs().sum(0) @ B.double().abs().t()
    assert float(((got - want).abs() / scale).max()) < 1e-6
    assert float((C.double().sum(0) - got).abs().max() / scale.max()) < 1e-6
    sq = (C.double() ** 2).sum(0)
    assert float(((S[:, 1].double().sum(0) - sq).abs() / sq).max()) < 1e-6
    dy = torch.randn(M, N, device="cuda", generator=g)
    ws = torch.empty(256 * N * K, device="cuda")
    f = _lib.load().msr3d_wgrad_rows_split
    outs = []
    for lo, hi in ((0, M), (0, M // 2), (M // 2, M)):
        dW = torch.empty(N, K, device="cuda")
        assert f(hi - lo, N, K, p(dy[lo:hi]), N, p(A[lo:hi]), K, p(dW), K, 0, p(ws), ws.numel(), None, st) == 0
        outs.append(dW.double())
    scale = float(outs[0].abs().max())
    assert float((outs[0] - (outs[1] + outs[2])).abs().max()) / scale < 1e-5
    # and a sample of its entries against float64 over all rows
    want = dy[:, :4].double().t() @ A.double()
    assert float((outs[0][:4] - want).abs().max()) / scale < 1e-5
