"""GPU: msr3d_rows_linear_split -- C = A W^T + b for a few hundred rows and a long reduction on the bf16 matrix pipe at
fp32 accuracy (the encoder's `fc`, /root/reference/modules/layers/pointnet.py:52-63, and `obj_linear_projection`,
/root/reference/model/ose3d_situation.py:284-290) -- against float64, through the C ABI."""
import pytest
import torch

from msr3d_amd import hipops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(M, N, K, seed, bias=True, lda_pad=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(M, K + lda_pad, generator=g)[:, :K] * 3.0
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) if bias else None
    return x, w, b


@pytest.mark.parametrize("M,N,K,bias,pad", [(960, 768, 768, True, 0), (960, 256, 768, True, 0), (61, 32, 128, False, 0),
                                             (1, 64, 256, True, 0), (130, 96, 384, True, 8), (1920, 768, 768, True, 0)])
def test_rows_linear_split_matches_float64(M, N, K, bias, pad):
    x, w, b = _case(M, N, K, 3 + M + N, bias, pad)
    xd = x.to(DEV)
    if pad:
        buf = torch.zeros(M, K + pad, device=DEV)
        buf[:, :K] = xd
        xd = buf[:, :K]                                    # row stride K + pad
    pack = hipops.pack_split_weight(w.to(DEV))
    y = hipops.rows_linear_split(xd, pack, N, None if b is None else b.to(DEV))
    want = x.double() @ w.double().t() + (0 if b is None else b.double())
    err = float((y.cpu().double() - want).abs().max())
    scale = float(want.abs().max())
    assert err <= 2e-6 * scale, (err, scale)               # fp32 accuracy: the f32 GEMM's own error at K = 768 is ~1e-6
    # bit-reproducible, and a row's result does not depend on which rows share the launch
    y2 = hipops.rows_linear_split(xd, pack, N, None if b is None else b.to(DEV))
    assert torch.equal(y, y2)
    if M > 70:
        sub = xd[37:37 + 65].contiguous()
        ys = hipops.rows_linear_split(sub, pack, N, None if b is None else b.to(DEV))
        assert torch.equal(ys, y[37:37 + 65])


def test_rows_linear_split_agrees_with_split_pack_launch():
    """The torch packer (frozen weights) and msr3d_split_pack (trainable weights, every step) write the same operand."""
    from msr3d_amd.scene_blocks import WeightPacks
    from msr3d_amd import _lib
    w = torch.randn(256, 768, device=DEV)
    packs = WeightPacks(torch.device(DEV))
    buf = packs.add("w", w, 256, 768, False)
    packs.launch(_lib.current_stream_ptr(torch.device(DEV)))
    torch.cuda.synchronize()
    assert torch.equal(buf.view(torch.int16), hipops.pack_split_weight(w))


def test_rows_linear_split_rejects_bad_shapes():
    x = torch.zeros(8, 128, device=DEV)
    pack = hipops.pack_split_weight(torch.zeros(32, 128, device=DEV))
    with pytest.raises(RuntimeError):
        hipops.rows_linear_split(x, pack, 48)              # N % 32
    with pytest.raises(RuntimeError):
        hipops.rows_linear_split(x[:, :96], pack, 32)      # K % 128
