"""Fused set-abstraction path for the frozen PointNet++ encoder.

Host side of msr3d_sa_fps2 / msr3d_sa_level (include/msr3d_hip.h): decides when the
fused kernels apply, packs the per-layer parameters once (conv weight with the K order
the kernels use, zero-padded; eval-mode BN folded to scale/shift) and issues the four
launches + the `fc` GEMM.  Anything that does not match the shipped configuration falls
back to the composite path (still the HIP ops, never the CPU).
"""
import ctypes

import torch
import torch.nn.functional as F

from .. import _lib
from . import pointnet2_utils as pu

_LEVEL_DIMS = ([6, 64, 64, 128], [131, 128, 128, 256], [259, 256, 512, 768])
_KP0 = (16, 144, 272)
_NSAMPLE = 32


def _level_spec(sa):
    """(dims, conv/bn pairs) of a single-scale SA module or None."""
    if len(sa.mlps) != 1:
        return None
    pairs = sa.mlps[0].conv_bn_pairs()
    if len(pairs) != 3 or any(c is None or c.kernel_size != (1, 1) for c, _ in pairs):
        return None
    dims = [pairs[0][0].in_channels] + [c.out_channels for c, _ in pairs]
    return dims, pairs


def can_fuse(net, pts):
    if not getattr(net, "use_fused", True) or not pts.is_cuda or pts.dtype != torch.float32:
        return False
    if pts.dim() != 3 or pts.size(-1) != 6 or len(net.encoder) != 3:
        return False
    if torch.is_grad_enabled() and (pts.requires_grad or any(p.requires_grad for p in net.parameters())):
        return False          # training the backbone needs autograd: composite path
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d) and m.training:
            return False      # batch statistics: composite path
    sa1, sa2, sa3 = net.encoder
    for sa, want in zip(net.encoder, _LEVEL_DIMS):
        spec = _level_spec(sa)
        if spec is None or spec[0] != want:
            return False
    g1, g2, g3 = sa1.groupers[0], sa2.groupers[0], sa3.groupers[0]
    if not (isinstance(g1, pu.QueryAndGroup) and isinstance(g2, pu.QueryAndGroup)
            and isinstance(g3, pu.GroupAll)):
        return False
    if not (g1.use_xyz and g2.use_xyz and g3.use_xyz) or g1.normalize_xyz or g2.normalize_xyz:
        return False
    if g1.nsample != _NSAMPLE or g2.nsample != _NSAMPLE:
        return False
    if sa1.npoint is None or sa2.npoint != 16 or sa3.npoint is not None or sa1.npoint > 64:
        return False
    if pts.size(1) * 12 > 64 * 1024:      # level-1 cloud is staged in LDS
        return False
    return True


def _pack_rows(conv, bn, kp, feat_first):
    """(weight rows (N, kp) in the kernels' K order, zero-padded; scale (N); shift (N))."""
    w = conv.weight.detach().reshape(conv.out_channels, conv.in_channels).float()
    n, k = w.shape
    wp = w.new_zeros((n, kp))
    if feat_first:            # kernel K order: [features, xyz]; reference: [xyz, features]
        wp[:, :k - 3] = w[:, 3:]
        wp[:, k - 3:k] = w[:, :3]
    else:
        wp[:, :k] = w
    if bn is not None:
        scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
        shift = bn.bias.detach().float() - bn.running_mean.float() * scale
        if conv.bias is not None:
            shift = shift + conv.bias.detach().float() * scale
    else:
        scale = torch.ones(n, device=w.device)
        shift = conv.bias.detach().float() if conv.bias is not None else torch.zeros(n, device=w.device)
    return wp, scale, shift


def _pack_layer(conv, bn, kp, feat_first):
    wp, scale, shift = _pack_rows(conv, bn, kp, feat_first)
    n = wp.shape[0]
    # MFMA-fragment order (include/msr3d_hip.h): [slab s][column tile t][lane = 16 g + i][j] =
    # W[16 t + i][16 s + 4 g + j] -- one contiguous 1 KB block per (slab, tile)
    frag = wp.view(n // 16, 16, kp // 16, 4, 4).permute(2, 0, 3, 1, 4)      # (s, t, g, i, j)
    return torch.cat([frag.reshape(-1), scale, shift]).contiguous()


_KP0_SPLIT = (32, 160, 288)


def _pack_layer_split(conv, bn, kp, feat_first, kperm=False):
    """Parameters for msr3d_sa_level_split: the weight split exactly into three bf16 terms, packed in
    16x16x32 MFMA-fragment order, and the folded BN affine (fp32).
    kperm: the K axis of each 32-wide slab numbered the way the previous layer's accumulators lie in the
    lanes (level 1 keeps its activations in registers: csrc/sa_split.hip) -- fragment position (g, e) of
    slab s holds channel 32 s + 16 (e >> 2) + 4 g + (e & 3)."""
    flat = _pack_rows(conv, bn, kp, feat_first)
    wp, scale, shift = flat
    n = wp.shape[0]
    if kperm:
        k = torch.arange(kp, device=wp.device)
        s_, g_, e_ = k // 32, (k % 32) // 8, k % 8
        wp = wp[:, 32 * s_ + 16 * (e_ >> 2) + 4 * g_ + (e_ & 3)]
    w0 = wp.to(torch.bfloat16)
    r1 = wp - w0.float()
    w1 = r1.to(torch.bfloat16)
    w2 = (r1 - w1.float()).to(torch.bfloat16)
    planes = torch.stack([w0, w1, w2])                                   # (3, n, kp)
    frag = planes.view(3, n // 16, 16, kp // 32, 4, 8).permute(3, 1, 0, 4, 2, 5)   # (s, t, p, g, i, j)
    return frag.contiguous().reshape(-1), torch.cat([scale, shift]).contiguous()


def _pack_level2_first(conv, bn):
    """Level 2's first layer for sa2_split_kernel (round 3): the 128 feature columns as split planes (K = 128),
    the three coordinate columns as fp32 rows (128, 4) appended to the affine -- the kernel adds their
    contribution to the accumulators with fp32 FMAs instead of a fifth, 29/32-empty slab of MFMAs."""
    w = conv.weight.detach().reshape(conv.out_channels, conv.in_channels).float()      # K order [xyz, features]
    n = w.shape[0]
    feat = w[:, 3:].contiguous()
    w0 = feat.to(torch.bfloat16)
    r1 = feat - w0.float()
    w1 = r1.to(torch.bfloat16)
    w2 = (r1 - w1.float()).to(torch.bfloat16)
    planes = torch.stack([w0, w1, w2])
    kp = feat.shape[1]
    frag = planes.view(3, n // 16, 16, kp // 32, 4, 8).permute(3, 1, 0, 4, 2, 5)       # (s, t, p, g, i, j)
    _, scale, shift = _pack_rows(conv, bn, conv.in_channels, False)
    wxyz = torch.zeros((n, 4), device=w.device)
    wxyz[:, :3] = w[:, :3]
    return frag.contiguous().reshape(-1), torch.cat([scale, shift, wxyz.reshape(-1)]).contiguous()


def _state_key(net):
    return tuple((t.data_ptr(), t._version) for t in list(net.parameters()) + list(net.buffers()))


def get_plan(net):
    key = _state_key(net)
    plan = getattr(net, "_fused_plan", None)
    if plan is not None and plan["key"] == key:
        return plan
    levels = []
    for li, sa in enumerate(net.encoder):
        _, pairs = _level_spec(sa)
        packed = []
        for j, (conv, bn) in enumerate(pairs):
            kp = _KP0[li] if j == 0 else conv.in_channels
            packed.append(_pack_layer(conv, bn, kp, feat_first=(j == 0 and li > 0)))
        levels.append(packed)
    dims = [(ctypes.c_int * 4)(*d) for d in _LEVEL_DIMS]
    # level 2 (the dominant kernel) also in split form: three bf16 planes for the bf16 matrix pipe
    _, pairs2 = _level_spec(net.encoder[1])
    split2 = [_pack_layer_split(conv, bn, _KP0_SPLIT[1] if j == 0 else conv.in_channels, feat_first=(j == 0))
              for j, (conv, bn) in enumerate(pairs2)]
    split2[0] = _pack_level2_first(*pairs2[0])
    _, pairs1 = _level_spec(net.encoder[0])
    split1 = [_pack_layer_split(conv, bn, _KP0_SPLIT[0] if j == 0 else conv.in_channels, feat_first=False,
                                kperm=(j > 0)) for j, (conv, bn) in enumerate(pairs1)]
    _, pairs3 = _level_spec(net.encoder[2])
    split3 = [_pack_layer_split(conv, bn, _KP0_SPLIT[2] if j == 0 else conv.in_channels, feat_first=(j == 0))
              for j, (conv, bn) in enumerate(pairs3)]
    plan = {"key": key, "levels": levels, "dims": dims, "split1": split1, "split2": split2, "split3": split3}
    # `fc` on the split kernels too (msr3d_rows_linear_split): the frozen weight packed once per plan
    from .. import hipops
    n_out, k_in = net.fc.weight.shape
    if hipops.rows_linear_split_ok(0, n_out, k_in) and net.fc.weight.is_cuda:
        plan["fc_split"] = hipops.pack_split_weight(net.fc.weight)
    net._fused_plan = plan
    return plan


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


PAD_VALUE = 1.0       # the dataset wrapper's padding cloud (dataset_wrapper.py:156-158)

# Matrix-pipe arithmetic of the level-2 SharedMLP (the dominant kernel):
#   "split"  every fp32 operand split exactly into three bf16 terms, six bf16 MFMA products per product,
#            fp32 accumulate (csrc/sa_split.hip): error per product below one fp32 rounding
#   "f32"    f32-input MFMA (csrc/sa_fused.hip): exact fp32 fma chains
#   "split2" LABELLED reduced variant, never the default: two bf16 terms per operand, three products (x0 w0 + x0 w1 + x1 w0,
#            ~16 significant bits per product), the same kernels from libmsr3d_hip_split2.so -- for comparison with what the
#            reference's cuDNN convolutions compute on its own hardware (TF32, 10 bits, by default)
# MSR3D_SA_MMA=f32|split|split2 or set_sa_mma(); see DESIGN.md §4.1 for the measured accuracy.
import os as _os
_sa_mma = [_os.environ.get("MSR3D_SA_MMA", "split")]
# MSR3D_FPS_QUERY=0: furthest-point sampling and level 1's ball query as the two launches of round 3
_FPS_QUERY = _os.environ.get("MSR3D_FPS_QUERY", "1") != "0"
if _sa_mma[0] not in ("f32", "split", "split2"):
    raise ValueError("MSR3D_SA_MMA must be 'f32', 'split' or 'split2'")


# Distinct-row kernels (round 5): a neighbourhood's SharedMLP over the min(hits, nsample) DIFFERENT rows ball_query
# found instead of all nsample (the rest are copies of the first hit; max is idempotent: same bits).
# MSR3D_SA_ROWS=0 or set_sa_rows(False): the all-rows kernels of rounds 2-4.
_sa_rows = [_os.environ.get("MSR3D_SA_ROWS", "1") != "0"]
# the planning launches of levels 1 and 2 as one (msr3d_sa_plan12); MSR3D_SA_PLAN12=0: each level plans in its own call
_PLAN12 = _os.environ.get("MSR3D_SA_PLAN12", "1") != "0"
# ... and both of them inside the sampling launch (msr3d_sa_fps2_query_plan); MSR3D_SA_PLAN_IN_SAMPLING=0: msr3d_sa_plan12
_PLAN_IN_SAMPLING = _os.environ.get("MSR3D_SA_PLAN_IN_SAMPLING", "1") != "0"
# the encoder's `fc` on msr3d_rows_linear_split (bf16 x 3 split) instead of the f32-input MFMA panel kernel
_FC_SPLIT = _os.environ.get("MSR3D_FC_SPLIT", "1") != "0"


# level 3 dealt over the objects' flags (msr3d_sa_level3_tiles); MSR3D_SA3_TILES=0: four consecutive objects a workgroup
_sa3_tiles = [_os.environ.get("MSR3D_SA3_TILES", "1") != "0"]


_plan_ws = {}


def _plan_workspaces(lib, dev, st, b, m1):
    """The two planning workspaces of a pass, kept per (device, stream, shape): they are arguments of the pass's FIRST
    launch now, and two allocator calls in front of it are host time the GPU idles through in the un-pipelined schedule.
    (Per stream: passes on different streams may overlap; passes on one stream are ordered.)"""
    if torch.cuda.is_current_stream_capturing():      # (a capture's allocations live in its own pool: never kept)
        return (torch.empty((int(lib.msr3d_sa_level1_rows_ws_bytes(b, m1)),), dtype=torch.uint8, device=dev),
                torch.empty((int(lib.msr3d_sa_level2_rows_ws_bytes(b)),), dtype=torch.uint8, device=dev))
    key = (str(dev), int(st.value) if hasattr(st, "value") and st.value else 0, b, m1)
    ws = _plan_ws.get(key)
    if ws is None:
        if len(_plan_ws) > 16:
            _plan_ws.clear()
        ws = (torch.empty((int(lib.msr3d_sa_level1_rows_ws_bytes(b, m1)),), dtype=torch.uint8, device=dev),
              torch.empty((int(lib.msr3d_sa_level2_rows_ws_bytes(b)),), dtype=torch.uint8, device=dev))
        _plan_ws[key] = ws
    return ws


def set_sa_rows(on):
    prev, _sa_rows[0] = _sa_rows[0], bool(on)
    return prev


def set_sa3_tiles(on):
    prev, _sa3_tiles[0] = _sa3_tiles[0], bool(on)
    return prev


def set_sa_mma(name):
    if name not in ("f32", "split", "split2"):
        raise ValueError("sa mma must be 'f32', 'split' or 'split2'")
    prev, _sa_mma[0] = _sa_mma[0], name
    return prev


def padding_feature(net, n_points, device):
    """Encoder output of the padding cloud (every coordinate and colour = 1.0), computed once per
    weight version and point count: what every padded object slot encodes to."""
    plan = get_plan(net)
    cache = plan.setdefault("pad_feat", {})
    key = (int(n_points), str(device))
    if key not in cache:
        ones = torch.full((1, n_points, 6), PAD_VALUE, dtype=torch.float32, device=device)
        cache[key] = forward(net, ones)[0].clone()
    return cache[key]


def forward(net, pts, return_internals=False, valid=None, out=None):
    """pts (b, P, 6) f32 contiguous -> (b, 768).  Four kernel launches + one GEMM (msr3d_gemm_f32);
    `out` (b, 768) f32 contiguous, optional: the result is written there (a step's static buffer).
    valid (b,) bool, optional: objects marked False are PADDING slots holding the constant cloud;
    the kernels skip them and their rows receive `padding_feature` -- the same values the encoder
    would produce, without encoding the same cloud once per slot."""
    pts = pts.contiguous()
    b, n, _ = pts.shape
    dev = pts.device
    plan = get_plan(net)
    vmask = pad_feat = None
    if valid is not None:
        pad_feat = padding_feature(net, n, dev)
        vmask = valid.reshape(b).contiguous().view(torch.uint8)
    lib = _lib.load()
    split = _sa_mma[0] in ("split", "split2")
    slib = _lib.load_split2() if _sa_mma[0] == "split2" else lib          # whose SharedMLP kernels run
    sa1, sa2, _ = net.encoder
    m1, m2 = sa1.npoint, sa2.npoint
    new1 = torch.empty((b, m1, 3), dtype=torch.float32, device=dev)
    new2 = torch.empty((b, m2, 3), dtype=torch.float32, device=dev)
    feat1 = torch.empty((b, m1, 128), dtype=torch.float32, device=dev)
    feat2 = torch.empty((b, m2, 256), dtype=torch.float32, device=dev)
    pooled = torch.empty((b, 768), dtype=torch.float32, device=dev)
    ball1 = torch.empty((b, m1, _NSAMPLE), dtype=torch.int32, device=dev)   # level-1 workspace
    # objects whose cloud is one repeated point (padding slots), reported by the sampling launch: one row per level
    constant = torch.empty((b,), dtype=torch.uint8, device=dev) if (_sa_rows[0] and split) else None
    dbg = {}
    if return_internals:
        dbg = {"idx1": torch.empty((b, m1), dtype=torch.int32, device=dev),
               "idx2": torch.empty((b, m2), dtype=torch.int32, device=dev),
               "ball1": ball1,
               "ball2": torch.empty((b, m2, _NSAMPLE), dtype=torch.int32, device=dev)}
    with torch.cuda.device(dev):
        st = _lib.current_stream_ptr(dev)
        # FPS of both levels and, beside it in the same workgroups, level 1's ball query (one launch instead of two; the
        # query's 24 us sit under the FPS chain's 33); clouds the fused kernel does not take: the two launches
        r1 = float(sa1.groupers[0].radius)
        queried = _FPS_QUERY and 256 < n <= 1024 and m1 <= 64
        L = plan["levels"]
        rows1 = split and _sa_rows[0] and m1 <= 64 and b < (1 << 18)
        rows2 = split and _sa_rows[0] and m1 <= 64 and m2 <= 16
        planned = 1 if (rows1 and rows2 and _PLAN12) else 0
        ws1 = ws2 = None
        in_launch = False                 # both plans written by the sampling launch itself (msr3d_sa_fps2_query_plan)
        with _lib.kernel_timer("msr3d_sa_fps2"):
            if queried and planned and _PLAN_IN_SAMPLING and slib is lib and constant is not None:
                ws1, ws2 = _plan_workspaces(lib, dev, st, b, m1)
                rc = lib.msr3d_sa_fps2_query_plan(b, n, 6, m1, m2, _p(pts), _p(dbg.get("idx1")), _p(new1), _p(dbg.get("idx2")),
                                                  _p(new2), _p(vmask), ctypes.c_float(r1), _NSAMPLE, _p(ball1), _p(constant),
                                                  _p(ws1), ctypes.c_float(sa2.groupers[0].radius), _p(feat2),
                                                  _p(dbg.get("ball2")), _p(ws2), st)
                in_launch = rc == 0
                if rc not in (0, -22):                          # (MSR3D_EINVAL: not its shape -- the separate calls below)
                    _lib.check(rc, "msr3d_sa_fps2_query_plan")
            if queried and not in_launch:
                rc = lib.msr3d_sa_fps2_query_flags(b, n, 6, m1, m2, _p(pts), _p(dbg.get("idx1")), _p(new1),
                                                   _p(dbg.get("idx2")), _p(new2), _p(vmask), ctypes.c_float(r1), _NSAMPLE,
                                                   _p(ball1), _p(constant), st)
                if rc == -22:                                   # MSR3D_EINVAL: not a shape of the fused kernel
                    queried = False
            if not queried:
                rc = lib.msr3d_sa_fps2_flags(b, n, 6, m1, m2, _p(pts), _p(dbg.get("idx1")), _p(new1),
                                             _p(dbg.get("idx2")), _p(new2), _p(vmask), _p(constant), st)
        _lib.check(rc, "msr3d_sa_fps2")
        if queried:
            r1 = 0.0                                            # level 1: ball1 already holds the neighbour lists
        with _lib.kernel_timer("msr3d_sa_level1"):
            if rows1:
                S = plan["split1"]
                if r1 > 0.0:                                    # (not queried beside the sampling: the query's own launch)
                    _lib.check(lib.msr3d_ball_query(b, n, m1, ctypes.c_float(r1), _NSAMPLE, _p(new1), _p(pts[..., :3].contiguous()),
                                                    _p(ball1), st), "msr3d_ball_query")
                if ws1 is None:
                    ws1 = torch.empty((int(slib.msr3d_sa_level1_rows_ws_bytes(b, m1)),), dtype=torch.uint8, device=dev)
                if planned and not in_launch:
                    # both levels' planning launches as one (each reads only what the sampling launch wrote); timed with
                    # level 1 -- level 2's timer then holds its products' launch alone
                    ws2 = torch.empty((int(slib.msr3d_sa_level2_rows_ws_bytes(b)),), dtype=torch.uint8, device=dev)
                    _lib.check(slib.msr3d_sa_plan12(b, m1, _p(ball1), _p(ws1), m1, m2, ctypes.c_float(sa2.groupers[0].radius),
                                                    _p(new1), _p(new2), _p(feat2), _p(dbg.get("ball2")), _p(ws2), _p(vmask),
                                                    _p(constant), st), "msr3d_sa_plan12")
                rc = slib.msr3d_sa_level1_rows(b, n, m1, _p(pts), _p(new1), _p(ball1), _p(S[0][0]), _p(S[0][1]), _p(S[1][0]),
                                              _p(S[1][1]), _p(S[2][0]), _p(S[2][1]), _p(feat1), _p(vmask), _p(constant),
                                              _p(ws1), planned, st)
            elif split:
                S = plan["split1"]
                rc = slib.msr3d_sa_level_split(1, b, n, m1, ctypes.c_float(r1), _p(pts), _p(None),
                                              _p(new1), _p(S[0][0]), _p(S[0][1]), _p(S[1][0]), _p(S[1][1]), _p(S[2][0]),
                                              _p(S[2][1]), _p(feat1), _p(ball1), _p(vmask), st)
            else:
                rc = lib.msr3d_sa_level(1, b, n, m1, ctypes.c_float(r1), _p(pts),
                                        _p(None), _p(new1), plan["dims"][0], _p(L[0][0]), _p(L[0][1]),
                                        _p(L[0][2]), _p(feat1), _p(ball1), _p(vmask), st)
        _lib.check(rc, "msr3d_sa_level(1)")
        with _lib.kernel_timer("msr3d_sa_level2"):
            if rows2:
                S = plan["split2"]
                ws = ws2 if planned else torch.empty((int(slib.msr3d_sa_level2_rows_ws_bytes(b)),), dtype=torch.uint8, device=dev)
                rc = slib.msr3d_sa_level2_rows(b, m1, m2, ctypes.c_float(sa2.groupers[0].radius), _p(new1),
                                              _p(feat1), _p(new2), _p(S[0][0]), _p(S[0][1]), _p(S[1][0]),
                                              _p(S[1][1]), _p(S[2][0]), _p(S[2][1]), _p(feat2),
                                              _p(dbg.get("ball2")), _p(vmask), _p(constant), _p(ws), planned, st)
            elif split:
                S = plan["split2"]
                rc = slib.msr3d_sa_level_split(2, b, m1, m2, ctypes.c_float(sa2.groupers[0].radius), _p(new1),
                                              _p(feat1), _p(new2), _p(S[0][0]), _p(S[0][1]), _p(S[1][0]),
                                              _p(S[1][1]), _p(S[2][0]), _p(S[2][1]), _p(feat2),
                                              _p(dbg.get("ball2")), _p(vmask), st)
            else:
                rc = lib.msr3d_sa_level(2, b, m1, m2, ctypes.c_float(sa2.groupers[0].radius), _p(new1),
                                        _p(feat1), _p(new2), plan["dims"][1], _p(L[1][0]), _p(L[1][1]),
                                        _p(L[1][2]), _p(feat2), _p(dbg.get("ball2")), _p(vmask), st)
        _lib.check(rc, "msr3d_sa_level(2)")
        with _lib.kernel_timer("msr3d_sa_level3"):
            if split and m2 == 16 and slib is lib and constant is not None and _sa3_tiles[0]:
                # three real objects a workgroup, the constant (padding) objects one row each: chosen on the device
                # from the flags of the sampling launch (csrc/sa_split.hip::sa3_tiles_kernel); same bits
                S = plan["split3"]
                rc = lib.msr3d_sa_level3_tiles(b, _p(new2), _p(feat2), _p(S[0][0]), _p(S[0][1]), _p(S[1][0]), _p(S[1][1]),
                                               _p(S[2][0]), _p(S[2][1]), _p(pooled), _p(vmask), _p(constant), st)
            elif split and m2 == 16:
                S = plan["split3"]
                rc = slib.msr3d_sa_level_split(3, b, m2, 1, ctypes.c_float(0.0), _p(new2), _p(feat2), _p(None),
                                              _p(S[0][0]), _p(S[0][1]), _p(S[1][0]), _p(S[1][1]), _p(S[2][0]),
                                              _p(S[2][1]), _p(pooled), _p(None), _p(vmask), st)
            else:
                rc = lib.msr3d_sa_level(3, b, m2, 1, ctypes.c_float(0.0), _p(new2), _p(feat2),
                                        _p(None), plan["dims"][2], _p(L[2][0]), _p(L[2][1]),
                                        _p(L[2][2]), _p(pooled), _p(None), _p(vmask), st)
        _lib.check(rc, "msr3d_sa_level(3)")
    if vmask is not None:
        # skipped rows of `pooled` are uninitialised memory: neutralise them before the GEMM (a NaN
        # would be harmless -- rows are independent -- but Inf * 0 noise in debuggers is not)
        pooled = torch.where(valid.reshape(b, 1), pooled, torch.zeros((), device=dev))
    # `fc` (/root/reference/modules/layers/pointnet.py:52-63) on the f32-MFMA token GEMM: the vendor
    # library's heuristic kernel was the last foreign launch of the encoder
    fc = net.fc
    n_out, k_in = fc.weight.shape
    res = out if (out is not None and vmask is None) else torch.empty((b, n_out), dtype=torch.float32, device=dev)
    if res.shape != (b, n_out) or not res.is_contiguous() or res.dtype != torch.float32:
        raise ValueError("out must be a contiguous (b, %d) float32 tensor" % n_out)
    if _sa_mma[0] != "f32" and "fc_split" in plan and _FC_SPLIT:
        # the bf16 matrix pipe at fp32 accuracy, as the levels (round 5; MSR3D_FC_SPLIT=0: the f32-input MFMA launch below)
        from .. import hipops
        hipops.rows_linear_split(pooled, plan["fc_split"], n_out, fc.bias, out=res)
        return _fc_tail(res, out, vmask, valid, pad_feat, return_internals, dbg,
                        dict(new_xyz1=new1, new_xyz2=new2, feat1=feat1, feat2=feat2, pooled=pooled, constant=constant)
                        if return_internals else None)
    # one K run per tile on the panel kernel: no K-split, so an object's feature does not depend on
    # which other objects share the launch (and no atomics: bit-reproducible)
    arr = (_lib.GemmProblem * 1)()
    q = arr[0]
    q.a_kc, q.b_kc, q.M, q.N, q.K = 1, 1, b, n_out, k_in
    q.A, q.lda, q.B, q.ldb = pooled.data_ptr(), k_in, fc.weight.data_ptr(), k_in
    q.C, q.ldc, q.bias, q.beta, q.single_run = res.data_ptr(), n_out, fc.bias.data_ptr(), 0.0, 1
    with torch.cuda.device(dev):
        rc = lib.msr3d_gemm_multi_f32(1, arr, _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_gemm_multi_f32")
    if vmask is not None:
        res = torch.where(valid.reshape(b, 1), res, pad_feat)
        if out is not None:
            out.copy_(res)
            res = out
    out = res
    if return_internals:
        dbg.update(new_xyz1=new1, new_xyz2=new2, feat1=feat1, feat2=feat2, pooled=pooled, constant=constant)
        return out, dbg
    return out


def _fc_tail(res, out, vmask, valid, pad_feat, return_internals, dbg, internals):
    if vmask is not None:
        res = torch.where(valid.reshape(res.shape[0], 1), res, pad_feat)
        if out is not None:
            out.copy_(res)
            res = out
    if return_internals:
        dbg.update(**internals)
        return res, dbg
    return res


# per-row multiply-accumulates of the three levels' SharedMLPs (configs/msr3d.yaml:198-201)
_ROW_MACS = (6 * 64 + 64 * 64 + 64 * 128, 131 * 128 + 128 * 128 + 128 * 256, 259 * 256 + 256 * 512 + 512 * 768)


def row_statistics(net, pts):
    """What one encoder pass over pts (b, P, 6) multiplies, per level (bench.py's roofline leg; one untimed pass):
    nominal rows (every neighbourhood slot), distinct rows (what the result needs: the min(hits, nsample)
    different rows of a centre, one row per level for a constant cloud) and the rows the matrix pipe is handed by the
    distinct-row kernels (whole 16-row tiles), each with its FLOPs (2 x MAC, fp32-accurate products)."""
    with torch.no_grad():
        _, dbg = forward(net, pts, return_internals=True)
    b = pts.shape[0]
    const = dbg["constant"].bool() if dbg.get("constant") is not None else torch.zeros(b, dtype=torch.bool, device=pts.device)
    real = ~const
    out = {}
    for lvl, key in ((1, "ball1"), (2, "ball2")):
        ball = dbg[key]
        m, ns = ball.shape[1], ball.shape[2]
        d = 1 + (ball[:, :, 1:] != ball[:, :, :1]).sum(-1)              # distinct rows per centre (first hit repeated)
        dr = d[real]
        distinct = int(dr.sum()) + int(const.sum())
        if lvl == 1:      # a wave's 32 rows: one big centre, or two centres of <= 16 distinct rows; a constant object: one
            small = (dr <= 16).sum(-1)
            tasks = (m - small) + (small + 1) // 2
            pipe_rows = 32 * (int(tasks.sum()) + int(const.sum()))
        else:             # chunks of 64 rows, 16-row tiles
            R = dr.sum(-1)
            pipe_rows = int(((R // 64) * 64 + ((R % 64 + 15) // 16) * 16).sum()) + 16 * int(const.sum())
        out[lvl] = {"nominal_rows": b * m * ns, "distinct_rows": distinct, "pipe_rows": pipe_rows}
    n3 = dbg["feat2"].shape[1]
    R, C = int(real.sum()), int(const.sum())
    pipe3 = b * n3
    if dbg.get("constant") is not None and _sa3_tiles[0] and _sa_mma[0] == "split":
        cus = torch.cuda.get_device_properties(pts.device).multi_processor_count
        grid = max((b + 3) // 4, min(cus, (b + 2) // 3 + (b + 47) // 48))
        if (R + 2) // 3 + (C + 47) // 48 <= grid:          # msr3d_sa_level3_tiles' device-side choice: 48-row tiles
            pipe3 = 48 * ((R + 2) // 3 + (C + 47) // 48)
    out[3] = {"nominal_rows": b * n3, "distinct_rows": n3 * R + C, "pipe_rows": pipe3}
    for lvl in (1, 2, 3):
        for k in ("nominal", "distinct", "pipe"):
            out[lvl][k + "_flop"] = 2.0 * _ROW_MACS[lvl - 1] * out[lvl][k + "_rows"]
    out["constant_objects"] = int(const.sum())
    out["objects"] = b
    return out
