"""msr3d_seq_ce_fwd / _bwd against the reference's own statements (model/msr3d/msr3d.py:426-441)
evaluated in float64: per-sequence mean cross-entropy over the supervised tokens, for f32 / bf16 /
f16 logits, ragged supervision (prompt positions -100), a sequence with a single target, vocabulary
sizes with and without a vector-width remainder, plus a roofline line at the LLM's shape."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def reference(logits, targets):
    """msr3d.py:426-441, in float64 on the values the kernel reads."""
    bs = logits.shape[0]
    lg = logits.double()
    shift_logits = lg[..., :-1, :].contiguous()
    shift_labels = targets[..., 1:].contiguous()
    n = (shift_labels >= 0).int().sum(1)
    loss = F.cross_entropy(shift_logits.view(-1, lg.shape[-1]), shift_labels.view(-1), reduction="none")
    return loss.view(bs, -1).sum(1) / n


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-6), (torch.float16, 2e-6)])
@pytest.mark.parametrize("B,T,V", [(4, 97, 32000), (3, 40, 32008), (2, 5, 1000), (1, 2, 264)])
def test_forward_and_backward_match_the_reference_statements(dtype, tol, B, T, V):
    from msr3d_amd.llm import seq_mean_cross_entropy
    torch.manual_seed(B * T + V)
    logits = (torch.randn(B, T, V, device="cuda") * 3).to(dtype).requires_grad_(True)
    targets = torch.randint(0, V, (B, T), device="cuda")
    targets[:, : T // 2] = -100                      # prompt / scene positions are not supervised
    if B > 1:
        targets[1, :] = -100
        targets[1, T - 1] = 7                        # a sequence with ONE supervised token
    loss = seq_mean_cross_entropy(logits, targets)
    ref_in = logits.detach().double().requires_grad_(True)      # the same (rounded) values, float64 math
    want = reference(ref_in, targets)
    assert torch.allclose(loss.double(), want, rtol=tol, atol=tol * 10), (loss, want)
    g = torch.rand(B, device="cuda") + 0.5
    loss.backward(g)
    want.backward(g.double())
    got, ref = logits.grad.double(), ref_in.grad
    # the gradient is written in the logits' dtype: tolerance = that dtype's rounding
    eps = {torch.float32: 1e-6, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    assert float((got - ref).norm() / ref.norm()) < 2 * eps
    assert got[:, -1].abs().max() == 0               # the last position has no target
    if T // 2 - 1 > 0:
        assert got[:, : T // 2 - 1].abs().max() == 0     # rows whose NEXT token is unsupervised


def test_sequence_without_targets_is_nan_like_the_reference():
    from msr3d_amd.llm import seq_mean_cross_entropy
    logits = torch.randn(2, 6, 512, device="cuda")
    targets = torch.full((2, 6), -100, device="cuda")
    targets[0, 3] = 5
    loss = seq_mean_cross_entropy(logits, targets)
    assert torch.isfinite(loss[0]) and torch.isnan(loss[1])


def test_cpu_path_is_the_reference_formulation():
    from msr3d_amd.llm import seq_mean_cross_entropy
    logits = torch.randn(2, 6, 50)
    targets = torch.randint(0, 50, (2, 6))
    targets[:, :2] = -100
    assert torch.allclose(seq_mean_cross_entropy(logits, targets), reference(logits, targets).float(), atol=1e-6)


def test_roofline_line_at_the_llm_shape(capsys):
    """4 sequences x 576 tokens x 32000 (Vicuna vocabulary), bf16: forward reads the logits once,
    backward reads them and writes the gradient: report GB/s against the 8 TB/s HBM peak."""
    from msr3d_amd.llm import seq_mean_cross_entropy
    B, T, V = 4, 576, 32000
    logits = torch.randn(B, T, V, device="cuda", dtype=torch.bfloat16).requires_grad_(True)
    targets = torch.randint(0, V, (B, T), device="cuda")
    g = torch.ones(B, device="cuda")
    for _ in range(3):
        seq_mean_cross_entropy(logits, targets).backward(g)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 20
    fw = bw = 0.0
    for _ in range(n):
        ev[0].record()
        loss = seq_mean_cross_entropy(logits, targets)
        ev[1].record()
        loss.backward(g)
        ev[2].record()
        torch.cuda.synchronize()
        fw += ev[0].elapsed_time(ev[1]) / n
        bw += ev[1].elapsed_time(ev[2]) / n
    byt = B * (T - 1) * V * 2
    # the kernels alone: 20 back-to-back C-ABI calls between two events (the autograd wrapper above also pays the
    # launches' host latency and torch's accumulation of the gradient into logits.grad)
    import ctypes
    from msr3d_amd import _lib
    lib, st = _lib.load(), _lib.current_stream_ptr(logits.device)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    lse = torch.empty(B, T - 1, device="cuda")
    tok = torch.empty(B, T - 1, device="cuda")
    loss_b = torch.empty(B, device="cuda")
    cnt = torch.empty(B, dtype=torch.int32, device="cuda")
    dlog = torch.empty_like(logits)
    x = logits.detach()

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20
    kf = timed(lambda: _lib.check(lib.msr3d_seq_ce_fwd(B, T, V, vp(x), 2, vp(targets), vp(lse), vp(tok), vp(loss_b), vp(cnt), st),
                                  "msr3d_seq_ce_fwd"))
    kb = timed(lambda: _lib.check(lib.msr3d_seq_ce_bwd(B, T, V, vp(x), 2, vp(targets), vp(lse), vp(cnt), vp(g), vp(dlog), st),
                                  "msr3d_seq_ce_bwd"))
    with capsys.disabled():
        print(f"\n[seq_ce] autograd wrapper: fwd {fw*1e3:.1f} us, bwd {bw*1e3:.1f} us; kernels alone: "
              f"fwd {kf*1e3:.1f} us = {byt/kf/1e6:.0f} GB/s ({byt/kf/1e6/6300:.2f} of the 6.3 TB/s a copy reaches), "
              f"bwd {kb*1e3:.1f} us = {2*byt/kb/1e6:.0f} GB/s ({2*byt/kb/1e6/6300:.2f})")
    assert torch.allclose(loss_b, seq_mean_cross_entropy(x, targets), atol=1e-5)
    assert fw > 0 and bw > 0
