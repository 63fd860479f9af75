"""The whole trainable part of the hot path -- obj_linear_projection, constant token embeddings,
positional encoders, three TransformerSpatialEncoderLayers, llm_proj -- as ONE autograd node
running a fixed schedule of fused launches, forward and backward
(/root/reference/model/ose3d_situation.py:284-439, modules/layers/transformers.py:200-252,314-329,
model/msr3d/msr3d.py:84-86,277).

Why a schedule and not a persistent kernel: on this part an in-kernel grid barrier costs 4-7 us
(MI355X_MICROARCH.md price list, barrier-xcd) against ~2 us for a kernel boundary
(tools/probe/launch_floor.hip: 1.7-2.3 us per dependent kernel inside a graph), so the chain is cut
at every all-to-all seam (projection -> attention -> projection -> FFN), and everything row-local is
folded into the GEMM that consumes it (csrc/strip_gemm.hip): per layer 5 launches forward and
5 backward instead of 10 + 9, 41 for the whole trainable forward + backward instead of 88.

    forward   step_begin | proj | pos | 3 x [LN+pos -> qkvc | attention | fc | LN,LN -> linear1+GELU | linear2] | LN -> llm_proj
    backward  llm_proj (dx, dW) | 3 x [LN-bwd -> dW2^T.. -> GELU-bwd | dx1, dW2, dW1 | LN,LN-bwd -> d_fc Wfc
              | attention-bwd | dx_qkvc, dW_qkvc, dW_fc] | pos-bwd | dW_loc, dW_size, dW_proj

All activations live in one arena allocated once per (batch, tokens) shape: addresses are fixed, so
the schedule is graph-capturable and allocates nothing per step.  Split-K meeting points share one
zero region filled by the first launch (which also bumps the dropout seed).  Weight gradients go
straight into the flat gradient buffer of the data-parallel engine (msr3d_amd/dp.py).

Numerics: the same arithmetic as the modular path (rowmath.h restates rowops.hip's row kernels; the
products are f32-input MFMA chains), dropout masks keyed by the same (seed, salt, element index), so
with equal salts the two paths draw identical masks.
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib, hipops, scene_blocks
from ._lib import BLK, EPI, PRO, GemmProblem, StripGemm

# "blocks": scene-local fused blocks on the bf16x3 matrix pipe (csrc/scene_block.hip, round 3);
# "strips": round 2's schedule of strip / panel / multi GEMM launches on the f32-input MFMA
_MODE = [os.environ.get("MSR3D_TRAINABLE", "blocks")]
if _MODE[0] not in ("blocks", "strips"):
    raise ValueError("MSR3D_TRAINABLE must be 'blocks' or 'strips'")


def set_mode(name):
    if name not in ("blocks", "strips"):
        raise ValueError("trainable-part mode must be 'blocks' or 'strips'")
    _MODE[0] = name

# MSR3D_MERGE_LAUNCHES=0: the step's zero fill and the LayerNorm-gradient column sums as launches of their own (round 3)
_MERGE = os.environ.get("MSR3D_MERGE_LAUNCHES", "1") != "0"
# MSR3D_PACK_FORK=1 (round 6, measured, default off unless it pays: DESIGN.md 4.2d): the step's weight split / pack launch
# (24 us, needed first by the first attention block) on a forked stream beside the launches that do not read the packs --
# the object projection, the positional encoders, the first rows launch (~30 us): one fork and one join in the graph
_PACK_FORK = os.environ.get("MSR3D_PACK_FORK", "0") == "1"

_vp = ctypes.c_void_p


def _ptr(t, offset_floats=0):
    if t is None:
        return None
    return _vp(t.data_ptr() + 4 * offset_floats)


class _Arena:
    """Bump allocator over one fp32 tensor; `zero=True` allocations are contiguous at the front."""

    def __init__(self, device):
        self.device = device
        self.specs = []          # (name, numel, zero)
        self.buf = None
        self.views = {}

    def want(self, name, *shape, zero=False):
        n = 1
        for s in shape:
            n *= s
        self.specs.append((name, (n + 3) // 4 * 4, zero, tuple(shape)))

    def build(self):
        order = [s for s in self.specs if s[2]] + [s for s in self.specs if not s[2]]
        total = sum(s[1] for s in order)
        self.buf = torch.zeros(total, dtype=torch.float32, device=self.device)
        off = 0
        for name, n, zero, shape in order:
            k = 1
            for s in shape:
                k *= s
            self.views[name] = self.buf[off:off + k].view(shape)
            off += n
        self.zero_floats = sum(s[1] for s in order if s[2])

    def __getitem__(self, k):
        return self.views[k]


def _direct(dp, params):
    return all(p is not None and getattr(p, "_msr3d_dp", None) is dp and p.is_leaf and p.grad is not None
               and p.is_contiguous() for p in params)


class PrompterSchedule:
    """Built by `attach(model, dp)` once the data-parallel engine and the flat optimiser own the
    parameter storage; `MSR3DHotPath.forward` routes through it while `eligible()` holds."""

    def __init__(self, model, dp):
        self.model, self.dp = model, dp
        self.pr = model.visual_prompter
        self.arena = None
        self.shape = None
        self.enabled = True
        # the step's first launch can also advance the dropout seed word (HotPathTrainStep sets this
        # and drops its own msr3d_bump_seed launch); off: the caller owns the seed
        self.bump_seed = False
        self.need_d_embeds = False   # set per backward: the object features come from an unfrozen encoder
        self.hybrid, self.tiles = False, 0
        self._pack_stream = None
        self._ln_job_buf = None

    # ------------------------------------------------------------------ eligibility
    def eligible(self, d, ignore_grad_mode=False):
        pr, cfg = self.pr, self.pr.cfg
        se = cfg.spatial_encoder
        e = d.get("obj_embeds")
        if not (self.enabled and (ignore_grad_mode or torch.is_grad_enabled()) and self.model.training and e is not None
                and e.is_cuda and e.dtype == torch.float32):
            return False
        if hipops._deterministic[0]:
            # bit-reproducible mode (ordered split-K, ordered LayerNorm-gradient sums) lives in the
            # per-layer path; this schedule's K-splits and column sums meet by float atomics
            return False
        if not (pr.situation_type in ("as_transform_for_objects", "as_object") and cfg.use_spatial_attn
                and se.obj_loc_encoding in ("same_all", "same_0") and se.pairwise_rel_type == "center"
                and se.spatial_dist_norm and se.spatial_dim == 5 and cfg.hidden_size == 256
                and se.spatial_attn_fusion == "cond" and se.activation == "gelu"
                and "single_obj" not in d):
            return False
        B, L = e.shape[:2]
        if L > 128 or e.shape[-1] % 4 or cfg.loc_fourier_dim > 64:
            return False
        if pr.situation_type == "as_object":
            # the agent as a token of its own (round 6): frozen encoder only
            # (a scene of more than 64 tokens -- BASELINE's stress configuration, L = 121 -- takes the strip schedule)
            if not (self.anchor and not e.requires_grad and L + 1 <= 128
                    and d.get("anchor_locs") is not None and d.get("anchor_orientation") is not None
                    and (not pr.use_orientation or pr.orientation_encoder.in_features == 84)):   # 4 + 8 x 10 bands
                return False
        # msr3d_pos_embed_bwd sums the positional gradient of at most three layers; the mask is read as bytes
        if not 1 <= len(pr.spatial_encoder) <= 3:
            return False
        msk = d.get("obj_masks")
        if msk is not None and msk.dtype != torch.bool:
            return False
        for layer in pr.spatial_encoder:
            sa = layer.self_attn
            if getattr(sa, "_packed", None) is None or sa.n_head * 32 != 256 or not sa.spatial_multihead:
                return False
            if layer.linear1.out_features % 64 or not getattr(layer, "use_fused_layer", True):
                return False
        return _direct(self.dp, self._params())

    @property
    def anchor(self):
        """situation_type 'as_object': the agent is a token in front of the scene's objects (L = O + 1)."""
        return self.pr.situation_type == "as_object" and bool(self.pr.use_anchor)

    def _params(self):
        pr, m = self.pr, self.model
        if self.anchor:
            ll = pr.loc_layers[0]
            ps = [pr.obj_linear_projection.weight, pr.obj_linear_projection.bias, pr.object_type_embedding.weight,
                  ll[0].weight, ll[0].bias, ll[1].weight, ll[1].bias, pr.anchor_feat, m.llm_proj.weight, m.llm_proj.bias]
            if pr.use_orientation:
                ps += [pr.orientation_encoder.weight, pr.orientation_encoder.bias]
        else:
            ps = [pr.obj_linear_projection.weight, pr.obj_linear_projection.bias, pr.object_type_embedding.weight,
                  pr.loc_embedding_encoder[0].weight, pr.loc_embedding_encoder[0].bias,
                  pr.loc_embedding_encoder[1].weight, pr.loc_embedding_encoder[1].bias,
                  pr.size_embedding_encoder[0].weight, pr.size_embedding_encoder[0].bias,
                  pr.size_embedding_encoder[1].weight, pr.size_embedding_encoder[1].bias,
                  m.llm_proj.weight, m.llm_proj.bias]
        if pr.use_orientation:
            ps.append(pr.object_orientation_feat)
        for layer in pr.spatial_encoder:
            sa = layer.self_attn
            ps += list(sa._packed_members) + [sa.fc.weight, sa.fc.bias, sa.layer_norm.weight, sa.layer_norm.bias,
                                              layer.norm1.weight, layer.norm1.bias, layer.norm2.weight,
                                              layer.norm2.bias, layer.linear1.weight, layer.linear1.bias,
                                              layer.linear2.weight, layer.linear2.bias]
        return ps

    # ------------------------------------------------------------------ storage
    def _ensure(self, B, L, KE, device):
        key = (B, L, KE, str(device), self.anchor)
        if self.shape == key:
            # The block tables hold RAW addresses of every parameter, every .grad view and the arena.  Anything
            # that re-points parameter storage after the tables were built -- FlatAdamW constructed later, a new
            # gradient engine, .to(), load_state_dict(assign=True) -- must rebuild them, or the blocks would pack
            # stale weights and write dW into old memory.
            if self.packs is not None and self._ptr_key != self._pointer_key():
                self._ln_job_buf = None
                self._build_block_tables()
            return
        pr, m = self.pr, self.model
        LO = L                                  # objects per scene; L from here on = TOKENS per scene
        L = LO + (1 if self.anchor else 0)
        D, M = 256, B * L
        nl = len(pr.spatial_encoder)
        sa0 = pr.spatial_encoder[0].self_attn
        W = sa0._packed[0].shape[0]
        H = sa0.n_head
        FF = pr.spatial_encoder[0].linear1.out_features
        E = m.llm_proj.out_features
        KF = 63 if self.anchor else pr.loc_embedding_encoder[0].in_features    # (anchor: the prologue's Fourier rows, unread)
        a = _Arena(device)
        a.want("x0", M, D, zero=True)
        if self.anchor:
            QF = pr.orientation_encoder.in_features if pr.use_orientation else 4
            a.want("a_ori", B, D, zero=True)     # orientation_encoder(fourier(quaternion)): one row per scene
            a.want("e61", M, KE)                 # the object features with a ZERO row in front of every scene (never written)
            a.want("qf", B, QF)                  # fourier(quaternion) rows

        for i in range(nl):
            a.want(f"ffn{i}", M, D, zero=True)
        a.want("d_tok", M, D, zero=True)
        blocks = self._blocks_capable(L, FF, H)
        if blocks:      # scene blocks: per-layer sums and dy's (kept for the deferred dW launch), partial slabs
            for i in range(nl):
                a.want(f"fcacc{i}", M, D); a.want(f"d_t{i}", M, D); a.want(f"d_xacc{i}", M, D)
                a.want(f"d_ffn{i}", M, D); a.want(f"d_pre{i}", M, FF); a.want(f"d_fc{i}", M, D)
                a.want(f"d_qkvc{i}", M, W)
            a.want("part", 20, M, D)       # a block's partial products, one slab per slice
            # LayerNorm parameter gradients: per-workgroup column sums of the backward row kernels (4 rows each),
            # added up in order by ONE launch at the end of backward (instead of 240 x 256 x 18 float atomics)
            a.want("lnpart", nl * 6, (M + 3) // 4, D)
            a.want("res", M, D)            # the residual gradient that joins the next sum
        a.want("loc6", M, 6)
        a.want("ff", M, KF)
        a.want("pw", B, L, L, 5)
        a.want("pos", M, D); a.want("sa", M, D); a.want("sta", M, 2); a.want("sb", M, D); a.want("stb", M, 2)
        for i in range(nl):
            a.want(f"xin{i}", M, D); a.want(f"qkvc{i}", M, W); a.want(f"probs{i}", B, H, L, L)
            a.want(f"ctx{i}", M, D); a.want(f"fc{i}", M, D)
            a.want(f"s1_{i}", M, D); a.want(f"st1_{i}", M, 2); a.want(f"s2_{i}", M, D); a.want(f"st2_{i}", M, 2)
            a.want(f"t{i}", M, D); a.want(f"pre{i}", M, FF); a.want(f"h{i}", M, FF)
            a.want(f"s3_{i}", M, D); a.want(f"st3_{i}", M, 2)
            a.want(f"d_xin{i}", M, D)
        a.want("tok", M, D)
        a.want("scene", M, E)
        # backward temporaries (one layer's worth, reused)
        a.want("d_ffn", M, D); a.want("d_t", M, D); a.want("d_pre", M, FF); a.want("d_fc", M, D)
        a.want("d_ctx", M, D); a.want("d_qkvc", M, W); a.want("d_la", M, D); a.want("d_lb", M, D)
        a.want("d_emb", M, KE)        # gradient of the object features: only written for an unfrozen encoder
        a.build()
        self.arena = a
        self._ln_job_buf = None
        self.pad = torch.zeros(M, dtype=torch.uint8, device=device)
        self.valid = torch.zeros((B, L), dtype=torch.bool, device=device)      # static copy of obj_masks
        self.freqs = torch.linspace(1.0, 15, steps=10, device=device)
        self.dims = dict(B=B, L=L, LO=LO, M=M, D=D, W=W, H=H, FF=FF, E=E, KF=KF, KE=KE, nl=nl)

        self.shape = key
        self.staged_for = None
        self.packs = self.wgrad = None
        self._ptr_key = None
        if blocks:
            # the blocks' input rows as three bf16 planes per scene; rows past L are never written (stay zero)
            self.hybrid = L > 64
            self.tiles = (M + 63) // 64 if self.hybrid else B         # 64-row tiles of the token matrix | scenes
            self.xp = torch.zeros(self.tiles * 3 * 64 * 256, dtype=torch.int16, device=device)
            self._build_block_tables()


    # ------------------------------------------------------------------ scene-local fused blocks (round 3)
    def _blocks_capable(self, L, FF, H):
        # (the arena's `part` has 20 partial slabs and msr3d_scene_block rejects more slices than that).  L <= 64: a scene is
        # a block.  64 < L <= 128 (round 6, BASELINE's stress configuration): the HYBRID schedule -- the row-local halves
        # (feed-forward, projector, every rows launch, the weight gradients) on the block kernels over 64-row tiles that
        # ignore scene boundaries (msr3d_scene_block_t.rows_total), the attention on the strip kernels
        return L <= 128 and FF % 128 == 0 and FF // 128 <= 16 and H == 8

    def use_blocks(self):
        # (the blocks' attention core multiplies on the bf16x3 split = fp32 accuracy; a reduced-precision attention
        # mode, hipops.set_attention_mma("bf16"), is honoured by the strip schedule's attention kernels)
        # (... and by the HYBRID schedule, whose attention half IS those kernels: scenes of more than 64 tokens)
        return _MODE[0] == "blocks" and self.packs is not None and (hipops._attn_mma[0] == "f32" or self.hybrid)

    def _pointer_key(self):
        """Addresses the block tables captured: every parameter and its gradient view."""
        return tuple((p.data_ptr(), p.grad.data_ptr() if p.grad is not None else 0) for p in self._params())

    def _build_block_tables(self):
        """Pack jobs for every weight operand of the blocks and the weight-gradient problem table; all
        sources / destinations are views of the flat parameter and gradient buffers or of the arena."""
        self._ptr_key = self._pointer_key()
        pr, m, a, dm = self.pr, self.model, self.arena, self.dims
        M, D, W, H, FF, E, KF, KE, nl = (dm[k] for k in ("M", "D", "W", "H", "FF", "E", "KF", "KE", "nl"))
        dev = a.buf.device
        pk = scene_blocks.WeightPacks(dev)
        wg = scene_blocks.WgradTable(dev)
        PIECE = scene_blocks.PIECE
        lp = m.llm_proj
        self.llm_blocks = E % 256 == 0 and E // 256 <= 20      # (20 partial slabs; wider projectors take the strip GEMM)
        if self.llm_blocks:
            pk.add("llm", lp.weight, E, D, False)                   # scene = tok W^T
            pk.add("llm_t", lp.weight, D, E, True)                  # d tok = d scene W
        self.wg_llm = wg.add(0, E, E, a["tok"].data_ptr(), D, D, M, lp.weight.grad.data_ptr(), D, lp.bias.grad.data_ptr())
        for i, layer in enumerate(pr.spatial_encoder):
            sa = layer.self_attn
            wv, bv, gwv, gbv, _dp = sa._packed
            l1, l2 = layer.linear1, layer.linear2
            if not self.hybrid:                         # (hybrid: the attention half reads the fp32 weights, strip kernels)
                head_rows = 8 * 8 * PIECE // 2              # int16 elements of one head's [8 slabs][8 tiles]
                buf = torch.empty(H * head_rows, dtype=torch.int16, device=dev)
                pk.bufs[f"qkvc{i}"] = buf
                buf_t = torch.empty(H * (4 * 16 * PIECE // 2), dtype=torch.int16, device=dev)
                pk.bufs[f"qkvc_t{i}"] = buf_t
                for h in range(H):
                    segs = scene_blocks.head_segments(h)
                    pk.add(None, wv, 128, D, False, segs, out=buf, out_offset=h * head_rows * 2)
                    pk.add(None, wv, D, 128, True, segs, out=buf_t, out_offset=h * 4 * 16 * PIECE)
                pk.add(f"fc{i}", sa.fc.weight, D, D, False)             # acc += ctx_h Wfc[:, h]^T
                pk.add(f"fc_t{i}", sa.fc.weight, D, D, True)            # d ctx = d_fc Wfc
            pk.add(f"w1_{i}", l1.weight, FF, D, False)              # pre = t W1^T
            pk.add(f"w1_t{i}", l1.weight, D, FF, True)              # d t += d_pre W1
            pk.add(f"w2_{i}", l2.weight, D, FF, False)              # ffn += h W2^T
            pk.add(f"w2_t{i}", l2.weight, FF, D, True)              # d h = d_ffn W2
            wg.add(a[f"d_ffn{i}"].data_ptr(), D, D, a[f"h{i}"].data_ptr(), FF, FF, M,
                   l2.weight.grad.data_ptr(), FF, l2.bias.grad.data_ptr())
            wg.add(a[f"d_pre{i}"].data_ptr(), FF, FF, a[f"t{i}"].data_ptr(), D, D, M,
                   l1.weight.grad.data_ptr(), D, l1.bias.grad.data_ptr())
            wg.add(a[f"d_qkvc{i}"].data_ptr(), W, W, a[f"xin{i}"].data_ptr(), D, D, M, gwv.data_ptr(), D, gbv.data_ptr())
            wg.add(a[f"d_fc{i}"].data_ptr(), D, D, a[f"ctx{i}"].data_ptr(), D, D, M,
                   sa.fc.weight.grad.data_ptr(), D, sa.fc.bias.grad.data_ptr())
        lpj = pr.obj_linear_projection
        if self.anchor:
            L = dm["L"]
            ll = pr.loc_layers[0]
            wg.add(a["d_la"].data_ptr(), D, D, a["loc6"].data_ptr(), 6, 6, M, ll[0].weight.grad.data_ptr(), 6,
                   ll[0].bias.grad.data_ptr())
            if pr.use_orientation:
                # dy = the AGENT rows of d xin0 (one per scene: row stride L * 256), x = fourier(quaternion)
                oe = pr.orientation_encoder
                QF = oe.in_features
                wg.add(a["d_xacc0"].data_ptr(), L * D, D, a["qf"].data_ptr(), QF, QF, dm["B"], oe.weight.grad.data_ptr(), QF,
                       oe.bias.grad.data_ptr())
            # x = the features with a zero row per agent: those rows add nothing to dW; the bias gradient (object rows
            # only) comes from msr3d_anchor_front_bwd
            self.wg_proj = wg.add(a["d_xacc0"].data_ptr(), D, D, a["e61"].data_ptr(), KE, KE, M, lpj.weight.grad.data_ptr(),
                                  KE, 0)
        else:
            le, se = pr.loc_embedding_encoder, pr.size_embedding_encoder
            wg.add(a["d_la"].data_ptr(), D, D, a["ff"].data_ptr(), KF, KF, M, le[0].weight.grad.data_ptr(), KF,
                   le[0].bias.grad.data_ptr())
            wg.add(a["d_lb"].data_ptr(), D, D, a["loc6"].data_ptr() + 12, 6, 3, M, se[0].weight.grad.data_ptr(), 3,
                   se[0].bias.grad.data_ptr())
            self.wg_proj = wg.add(a["d_xacc0"].data_ptr(), D, D, 0, KE, KE, M, lpj.weight.grad.data_ptr(), KE,
                                  lpj.bias.grad.data_ptr())
        self.packs, self.wgrad = pk, wg

    def forward_blocks(self, embeds):
        """forward   step_begin | pack | proj | pos | 3 x [rows | attention block | rows | feed-forward block]
                     | rows | llm_proj        (rows = msr3d_scene_rows: sum of the partial slabs + the row-local
                     dropout / residual / LayerNorm chain, once per row, -> bf16 planes)"""
        pr, m, a, dm = self.pr, self.model, self.arena, self.dims
        B, L, M, D, W, H, FF, E, KF, KE, nl = (dm[k] for k in ("B", "L", "M", "D", "W", "H", "FF", "E", "KF", "KE", "nl"))
        dev = embeds.device
        self.lib = lib = _lib.load()
        self.stream = st = _lib.current_stream_ptr(dev)
        train = m.training
        seed = hipops.seed_word(dev)
        anchor = self.anchor
        if anchor:
            # object features behind a zero agent row per scene (one strided copy; the agent rows are never written)
            a["e61"].view(B, L, KE)[:, 1:].copy_(embeds.reshape(B, L - 1, KE))
            e2 = a["e61"]
        else:
            e2 = embeds.reshape(M, KE)
            e2 = e2 if e2.is_contiguous() else e2.contiguous()
        self.saved_embeds = e2
        layers = list(pr.spatial_encoder)
        self.salts = [[hipops._next_salt() for _ in range(4)] for _ in layers]
        self.ps = []
        for layer in layers:
            sa = layer.self_attn
            self.ps.append((float(sa.dropout.p) if train else 0.0, float(layer.dropout1.p) if train else 0.0,
                            float(layer.dropout2.p) if train else 0.0, float(layer.dropout.p) if train else 0.0))
        same_all = pr.cfg.spatial_encoder.obj_loc_encoding == "same_all"
        pk, xp, part, MD = self.packs, self.xp, a["part"], M * D
        hyb = self.hybrid
        # hybrid: Bt tiles of Lt = 64 rows over the whole token matrix instead of B scenes of L rows
        Bt, Lt, RT = (self.tiles, 64, M) if hyb else (B, L, 0)

        def blk(st_, **kw):
            scene_blocks.launch_block(st_, rows_total=RT, **kw)
        rows = scene_blocks.launch_rows
        nff = FF // 128
        join = None
        with torch.cuda.device(dev):
            if _PACK_FORK:
                rc = lib.msr3d_step_begin(_ptr(a.buf), a.zero_floats, _ptr(seed) if self.bump_seed else None, st)
                _lib.check(rc, "msr3d_step_begin")
                if self._pack_stream is None:
                    self._pack_stream = torch.cuda.Stream(device=dev)
                side = self._pack_stream
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    pk.launch(_lib.current_stream_ptr(dev))
                join = side
            elif _MERGE:
                # ONE launch: this step's weights split and fragment-packed + the arena's zero fill + the seed bump
                pk.launch(st, begin=(a.buf, a.zero_floats, seed if self.bump_seed else None))
            else:
                rc = lib.msr3d_step_begin(_ptr(a.buf), a.zero_floats, _ptr(seed) if self.bump_seed else None, st)
                _lib.check(rc, "msr3d_step_begin")
                pk.launch(st)
            lp = pr.obj_linear_projection
            front = [dict(a_kc=1, b_kc=1, M=M, N=D, K=KE, A=e2, lda=KE, B=lp.weight, ldb=KE, C=a["x0"], ldc=D,
                          bias=lp.bias, beta=1.0)]
            if anchor:
                if pr.use_orientation:       # the agent's orientation term, one row per scene, in the projection's launch
                    oe = pr.orientation_encoder
                    front.append(dict(a_kc=1, b_kc=1, M=B, N=D, K=oe.in_features, A=a["qf"], lda=oe.in_features, B=oe.weight,
                                      ldb=oe.in_features, C=a["a_ori"], ldc=D, bias=oe.bias, beta=1.0))
                self._multi(front)
                ll = pr.loc_layers[0]
                # agent / object rows assembled, LN(loc_layers[0]) added, -> xin0 and the first block's planes: one launch
                rc = lib.msr3d_anchor_front_fwd(B, L, _ptr(a["x0"]), _ptr(a["a_ori"]), _ptr(pr.anchor_feat),
                                                _ptr(pr.object_type_embedding.weight),
                                                _ptr(pr.object_orientation_feat) if pr.use_orientation else None,
                                                _ptr(a["loc6"]), _ptr(ll[0].weight), _ptr(ll[0].bias), _ptr(ll[1].weight),
                                                _ptr(ll[1].bias), ctypes.c_float(ll[1].eps), _ptr(a["pos"]), _ptr(a["sa"]),
                                                _ptr(a["sta"]), _ptr(a["xin0"]), None if hyb else _vp(xp.data_ptr()), st)
                _lib.check(rc, "msr3d_anchor_front_fwd")
            else:
                self._multi(front)
                le, se = pr.loc_embedding_encoder, pr.size_embedding_encoder
                # positional term + the layer input (tokens + positional term + the two constant rows) and its planes: one
                # launch (the MSR3D_PRO_ADD rows launch of rounds 3-5 rides in it)
                rc = lib.msr3d_pos_embed_tokens_fwd(
                    M, L, KF, _ptr(a["ff"]), _ptr(a["loc6"]), _ptr(le[0].weight), _ptr(le[0].bias), _ptr(le[1].weight),
                    _ptr(le[1].bias), ctypes.c_float(le[1].eps), _ptr(se[0].weight), _ptr(se[0].bias), _ptr(se[1].weight),
                    _ptr(se[1].bias), ctypes.c_float(se[1].eps), _ptr(a["pos"]), _ptr(a["sa"]), _ptr(a["sta"]), _ptr(a["sb"]),
                    _ptr(a["stb"]), _ptr(a["x0"]), _ptr(pr.object_type_embedding.weight),
                    _ptr(pr.object_orientation_feat) if pr.use_orientation else None, _ptr(a["xin0"]),
                    None if hyb else _vp(xp.data_ptr()), st)
                _lib.check(rc, "msr3d_pos_embed_tokens_fwd")
            for i, layer in enumerate(layers):
                sa = layer.self_attn
                bv = sa._packed[1]
                p_attn, p1, p2, p_ffn = self.ps[i]
                s_attn, s_1, s_2, s_ffn = self.salts[i]
                if i == 0:
                    pass        # (msr3d_pos_embed_tokens_fwd / msr3d_anchor_front_fwd wrote xin0 and its planes)
                else:           # previous layer's closing norm (+ the positional term)
                    prev = layers[i - 1]
                    rows(st, M=M, L=Lt, pro=PRO["ln"], part=part, nslab=nff, part_stride=MD, a0_bias=prev.linear2.bias,
                         sum_out=a[f"ffn{i-1}"], a1=a[f"t{i-1}"], a2=a["pos"] if same_all else None,
                         g1=prev.norm2.weight, b1=prev.norm2.bias, eps1=prev.norm2.eps, p1=self.ps[i - 1][2],
                         salt1=self.salts[i - 1][2], seed=seed, o0=a[f"s3_{i-1}"], ost1=a[f"st3_{i-1}"],
                         o1=a[f"xin{i}"], xp=None if hyb else xp)
                if join is not None:        # the packs are needed from here on
                    torch.cuda.current_stream(dev).wait_stream(join)
                    join = None
                if hyb:
                    # the attention half on the strip kernels (a scene of more than 64 tokens does not fit a block):
                    # q | k | v | cond projection, attention, out-projection (+ bias) -> fcacc
                    wv = sa._packed[0]
                    self._strip(M=M, N=W, pro=PRO["plain"], epi=EPI["bias"], b_kc=1, a0=a[f"xin{i}"], W=wv, ldw=D, bias=bv,
                                C=a[f"qkvc{i}"], ldc=W)
                    q = a[f"qkvc{i}"]
                    base, fs = q.data_ptr(), 4
                    rc = lib.msr3d_spatial_attn_fwd(B, L, H, D // H, 5, _vp(base), _vp(base + D * fs), _vp(base + 2 * D * fs),
                                                    W, _vp(base + 3 * D * fs), W, _ptr(a["pw"]), _ptr(self.pad),
                                                    _ptr(a[f"ctx{i}"]), _ptr(a[f"probs{i}"]), hipops.attention_mma(False, training=True), st)
                    _lib.check(rc, "msr3d_spatial_attn_fwd")
                    self._strip(M=M, N=D, pro=PRO["plain"], epi=EPI["bias"], b_kc=1, a0=a[f"ctx{i}"], W=sa.fc.weight,
                                ldw=D, bias=sa.fc.bias, C=a[f"fcacc{i}"], ldc=D)
                    rows(st, M=M, L=Lt, pro=PRO["ln2"], a0=a[f"fcacc{i}"], nslab=0, a1=a[f"xin{i}"],
                         g1=sa.layer_norm.weight, b1=sa.layer_norm.bias, eps1=sa.layer_norm.eps, p1=p_attn, salt1=s_attn,
                         g2=layer.norm1.weight, b2=layer.norm1.bias, eps2=layer.norm1.eps, p2=p1, salt2=s_1, seed=seed,
                         o0=a[f"s1_{i}"], ost1=a[f"st1_{i}"], o2=a[f"s2_{i}"], ost2=a[f"st2_{i}"], o1=a[f"t{i}"], xp=xp)
                else:
                    blk(st, kind=BLK["attn_fwd"], B=B, L=L, xp=xp, w1=pk.bufs[f"qkvc{i}"], w1_bytes=pk.nbytes(f"qkvc{i}"),
                        bias1=bv, w2=pk.bufs[f"fc{i}"], w2_bytes=pk.nbytes(f"fc{i}"), part=part, part_stride=MD,
                        qkvc=a[f"qkvc{i}"], ldq=W, ploc=a["pw"], pad=self.pad, probs=a[f"probs{i}"], ctx=a[f"ctx{i}"], H=H)
                    # attention tail's LayerNorm, then norm1 over the same residual -> the feed-forward block's input
                    rows(st, M=M, L=L, pro=PRO["ln2"], part=part, nslab=H, part_stride=MD, a0_bias=sa.fc.bias,
                         sum_out=a[f"fcacc{i}"], a1=a[f"xin{i}"], g1=sa.layer_norm.weight, b1=sa.layer_norm.bias,
                         eps1=sa.layer_norm.eps, p1=p_attn, salt1=s_attn, g2=layer.norm1.weight, b2=layer.norm1.bias,
                         eps2=layer.norm1.eps, p2=p1, salt2=s_1, seed=seed, o0=a[f"s1_{i}"], ost1=a[f"st1_{i}"],
                         o2=a[f"s2_{i}"], ost2=a[f"st2_{i}"], o1=a[f"t{i}"], xp=xp)
                blk(st, kind=BLK["ffn_fwd"], B=Bt, L=Lt, xp=xp, w1=pk.bufs[f"w1_{i}"], w1_bytes=pk.nbytes(f"w1_{i}"),
                    bias1=layer.linear1.bias, w2=pk.bufs[f"w2_{i}"], w2_bytes=pk.nbytes(f"w2_{i}"), part=part,
                    part_stride=MD, pre=a[f"pre{i}"], h=a[f"h{i}"], ff=FF, p_drop=p_ffn, salt=s_ffn, seed=seed)
            last = layers[-1]
            tail = dict(M=M, L=Lt, pro=PRO["ln"], part=part, nslab=nff, part_stride=MD, a0_bias=last.linear2.bias,
                        sum_out=a[f"ffn{nl-1}"], a1=a[f"t{nl-1}"], g1=last.norm2.weight, b1=last.norm2.bias,
                        eps1=last.norm2.eps, p1=self.ps[-1][2], salt1=self.salts[-1][2], seed=seed,
                        o0=a[f"s3_{nl-1}"], ost1=a[f"st3_{nl-1}"], o1=a["tok"])
            if self.llm_blocks:
                rows(st, xp=xp, **tail)
                blk(st, kind=BLK["linear"], B=Bt, L=Lt, xp=xp, w1=pk.bufs["llm"], w1_bytes=pk.nbytes("llm"),
                    bias1=m.llm_proj.bias, C=a["scene"], ldc=E, N=E)
            else:
                rows(st, **tail)
                self._strip(M=M, N=E, pro=PRO["plain"], epi=EPI["bias"], b_kc=1, a0=a["tok"], W=m.llm_proj.weight,
                            ldw=D, bias=m.llm_proj.bias, C=a["scene"], ldc=E)
        return a["tok"].view(B, L, D), a["scene"].view(B, L, E)

    def backward_blocks(self, g_scene, g_tok):
        """backward  d tok partials = d scene W_llm | 3 x [rows | feed-forward block bwd | rows | attention block
                     bwd] | rows (sum) | pos-bwd | ALL weight gradients in one launch"""
        pr, m, a, dm = self.pr, self.model, self.arena, self.dims
        B, L, M, D, W, H, FF, E, KF, KE, nl = (dm[k] for k in ("B", "L", "M", "D", "W", "H", "FF", "E", "KF", "KE", "nl"))
        dev = a.buf.device
        lib = self.lib
        self.stream = st = _lib.current_stream_ptr(dev)
        seed = hipops.seed_word(dev)
        layers = list(pr.spatial_encoder)
        same_all = pr.cfg.spatial_encoder.obj_loc_encoding == "same_all"
        pk, wg, xp, part, MD = self.packs, self.wgrad, self.xp, a["part"], M * D
        hyb = self.hybrid
        Bt, Lt, RT = (self.tiles, 64, M) if hyb else (B, L, 0)

        def blk(st_, **kw):
            scene_blocks.launch_block(st_, rows_total=RT, **kw)
        rows = scene_blocks.launch_rows
        nff = FF // 128
        with torch.cuda.device(dev):
            lp = m.llm_proj
            # the top layer's upstream gradient: d tok = d scene W_llm (+ a direct consumer of obj_tokens)
            src = dict(a0=a["d_tok"])                       # zeros (step_begin) unless something arrives
            if g_tok is not None:
                a["d_tok"].add_(g_tok.reshape(M, D))
            if g_scene is not None:
                g = g_scene.reshape(M, E)
                g = g if g.is_contiguous() else g.contiguous()
                self._g_keep = g
                if self.llm_blocks:
                    blk(st, kind=BLK["linear_ksplit"], B=Bt, L=Lt, a0=g, lda0=E, w1=pk.bufs["llm_t"],
                        w1_bytes=pk.nbytes("llm_t"), part=part, part_stride=MD)
                    src = dict(part=part, nslab=E // 256, part_stride=MD, extra=a["d_tok"] if g_tok is not None else None)
                else:
                    self._multi([dict(a_kc=1, b_kc=0, M=M, N=D, K=E, A=g, lda=E, B=lp.weight, ldb=D, C=a["d_tok"],
                                      ldc=D, beta=1.0)])
                wg.set_ptr(self.wg_llm, "dy", g.data_ptr())
                wg.set_ptr(self.wg_llm, "M", M)
            else:
                wg.set_ptr(self.wg_llm, "M", 0)             # no upstream gradient for llm_proj in this step
            for i in range(nl - 1, -1, -1):
                layer = layers[i]
                sa = layer.self_attn
                p_attn, p1, p2, p_ffn = self.ps[i]
                s_attn, s_1, s_2, s_ffn = self.salts[i]
                # d_out (sum) -> LN(norm2)-bwd: residual gradient -> res, dropout-bwd -> d_ffn (+ planes)
                lnp = a["lnpart"]
                rows(st, M=M, L=Lt, pro=PRO["lnbwd"], a1=a[f"s3_{i}"], st1=a[f"st3_{i}"], g1=layer.norm2.weight,
                     p1=p2, salt1=s_2, seed=seed, o0=a[f"d_ffn{i}"], o1=a["res"], dg1=lnp[6 * i], db1=lnp[6 * i + 1],
                     grad_partials=1, sum_out=a[f"d_xacc{i+1}"] if i + 1 < nl else None, xp=xp, **src)
                # d_h = d_ffn W2 -> GELU-bwd -> d_pre; partials of d_pre W1
                blk(st, kind=BLK["ffn_bwd"], B=Bt, L=Lt, xp=xp, w1=pk.bufs[f"w2_t{i}"], w1_bytes=pk.nbytes(f"w2_t{i}"),
                    w2=pk.bufs[f"w1_t{i}"], w2_bytes=pk.nbytes(f"w1_t{i}"), part=part, part_stride=MD,
                    pre=a[f"pre{i}"], h=a[f"d_pre{i}"], ff=FF, p_drop=p_ffn, salt=s_ffn, seed=seed)
                # d_t = residual + sum -> LN(norm1), LN(attention tail) bwd: residual -> res, d_fc (+ planes)
                rows(st, M=M, L=Lt, pro=PRO["ln2bwd"], part=part, nslab=nff, part_stride=MD, extra=a["res"],
                     sum_out=a[f"d_t{i}"], a1=a[f"s1_{i}"], a2=a[f"s2_{i}"], st1=a[f"st1_{i}"], st2=a[f"st2_{i}"],
                     g1=sa.layer_norm.weight, g2=layer.norm1.weight, p1=p_attn, salt1=s_attn, p2=p1, salt2=s_1,
                     seed=seed, o0=a[f"d_fc{i}"], o1=a[f"d_xin{i}"] if hyb else a["res"], dg1=lnp[6 * i + 2],
                     db1=lnp[6 * i + 3], dg2=lnp[6 * i + 4], db2=lnp[6 * i + 5], grad_partials=1,
                     xp=None if hyb else xp)
                if hyb:
                    # the attention half on the strip kernels: d ctx = d_fc Wfc | attention backward | d xin += d[q|k|v|cond] W
                    # (the residual gradient is already in d_xin)
                    wv = sa._packed[0]
                    self._strip(M=M, N=D, pro=PRO["plain"], epi=EPI["bias"], b_kc=0, a0=a[f"d_fc{i}"], W=sa.fc.weight, ldw=D,
                                C=a["d_ctx"], ldc=D)
                    q, gq = a[f"qkvc{i}"], a[f"d_qkvc{i}"]
                    base, gb, fs = q.data_ptr(), gq.data_ptr(), 4
                    rc = lib.msr3d_spatial_attn_bwd(B, L, H, D // H, 5, _vp(base), _vp(base + D * fs), _vp(base + 2 * D * fs),
                                                    W, _vp(base + 3 * D * fs), W, _ptr(a["pw"]), _ptr(self.pad),
                                                    _ptr(a[f"probs{i}"]), _ptr(a["d_ctx"]), _vp(gb), _vp(gb + D * fs),
                                                    _vp(gb + 2 * D * fs), W, _vp(gb + 3 * D * fs), W,
                                                    hipops.attention_mma(True), st)
                    _lib.check(rc, "msr3d_spatial_attn_bwd")
                    self._multi([dict(a_kc=1, b_kc=0, M=M, N=D, K=W, A=gq, lda=W, B=wv, ldb=D, C=a[f"d_xin{i}"], ldc=D,
                                      beta=1.0)])
                    src = dict(a0=a[f"d_xin{i}"])
                else:
                    # d_ctx = d_fc Wfc; attention bwd; partials of d[q|k|v|cond] W
                    blk(st, kind=BLK["attn_bwd"], B=B, L=L, xp=xp, w1=pk.bufs[f"fc_t{i}"], w1_bytes=pk.nbytes(f"fc_t{i}"),
                        w2=pk.bufs[f"qkvc_t{i}"], w2_bytes=pk.nbytes(f"qkvc_t{i}"), part=part, part_stride=MD,
                        qkvc=a[f"qkvc{i}"], ldq=W, dqkvc=a[f"d_qkvc{i}"], ploc=a["pw"], pad=self.pad,
                        probs=a[f"probs{i}"], H=H)
                    src = dict(part=part, nslab=H, part_stride=MD, extra=a["res"])
            rows(st, M=M, L=Lt, pro=PRO["plain"], sum_out=a["d_xacc0"], **src)     # d_xin0, whole
            more = same_all and nl > 1
            if self.anchor:
                ll = pr.loc_layers[0]
                tg = pr.object_type_embedding.weight.grad
                rc = lib.msr3d_anchor_front_bwd(
                    B, L, _ptr(a["d_xacc0"]), _ptr(a["d_xacc1"]) if more else None,
                    _ptr(a["d_xacc2"]) if (more and nl > 2) else None, _ptr(a["sa"]), _ptr(a["sta"]), _ptr(ll[1].weight),
                    _ptr(a["d_la"]), _ptr(ll[1].weight.grad), _ptr(ll[1].bias.grad), _ptr(tg),
                    _ptr(pr.object_orientation_feat.grad) if pr.use_orientation else None,
                    _ptr(pr.obj_linear_projection.bias.grad), _ptr(tg, D), _ptr(pr.anchor_feat.grad), st)
                _lib.check(rc, "msr3d_anchor_front_bwd")
            else:
                le, se = pr.loc_embedding_encoder, pr.size_embedding_encoder
                rc = lib.msr3d_pos_embed_bwd(
                    M, _ptr(a["d_xacc0"]), _ptr(a["d_xacc1"]) if more else None,
                    _ptr(a["d_xacc2"]) if (more and nl > 2) else None, _ptr(a["sa"]), _ptr(a["sta"]), _ptr(le[1].weight),
                    _ptr(a["sb"]), _ptr(a["stb"]), _ptr(se[1].weight), _ptr(a["d_la"]), _ptr(a["d_lb"]),
                    _ptr(le[1].weight.grad), _ptr(le[1].bias.grad), _ptr(se[1].weight.grad), _ptr(se[1].bias.grad),
                    _ptr(pr.object_type_embedding.weight.grad),
                    _ptr(pr.object_orientation_feat.grad) if pr.use_orientation else None, st)
                _lib.check(rc, "msr3d_pos_embed_bwd")
            wg.set_ptr(self.wg_proj, "x", self.saved_embeds.data_ptr())
            if _MERGE:
                # every weight gradient + (as extra workgroups) the ordered LayerNorm-gradient column sums: one launch
                wg.launch(st, colsum=(len(layers) * 6, self._ln_jobs(layers)))
            else:
                _lib.check(lib.msr3d_colsum_partials(len(layers) * 6, _ptr(self._ln_jobs(layers)), st), "msr3d_colsum_partials")
                wg.launch(st)
            if self.need_d_embeds:     # unfrozen object encoder: d obj_embeds = d_xin0 W_proj
                lpj = pr.obj_linear_projection
                self._multi([dict(a_kc=1, b_kc=0, M=M, N=KE, K=D, A=a["d_xacc0"], lda=D, B=lpj.weight, ldb=KE,
                                  C=a["d_emb"], ldc=KE, beta=0.0)])
        for p in self._params():
            self.dp.mark_ready(p)

    def _ln_jobs(self, layers):
        """Job table of msr3d_colsum_partials: (partial buffer, rows, gradient vector) per LayerNorm parameter."""
        if self._ln_job_buf is None:
            from ._lib import ColsumJob
            lnp, n = self.arena["lnpart"], (self.dims["M"] + 3) // 4
            jobs = []
            for i, layer in enumerate(layers):
                sa = layer.self_attn
                for k, g in enumerate((layer.norm2.weight.grad, layer.norm2.bias.grad, sa.layer_norm.weight.grad,
                                       sa.layer_norm.bias.grad, layer.norm1.weight.grad, layer.norm1.bias.grad)):
                    jobs.append(ColsumJob(part=lnp[6 * i + k].data_ptr(), dst=g.data_ptr(), n=n))
            arr = (ColsumJob * len(jobs))(*jobs)
            self._ln_job_buf = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(lnp.device)
        return self._ln_job_buf

    # ------------------------------------------------------------------ launch helpers
    def _strip(self, **kw):
        s = StripGemm()
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                v = v.data_ptr()
            elif isinstance(v, _vp):
                v = v.value
            setattr(s, k, v if v is not None else 0)
        rc = self.lib.msr3d_strip_gemm_f32(ctypes.byref(s), self.stream)
        if rc:
            fields = ", ".join(f"{n}={getattr(s, n)!r}" for n, _ in s._fields_ if getattr(s, n))
            _lib.check(rc, f"msr3d_strip_gemm_f32({fields})")

    def _multi(self, probs):
        arr = (GemmProblem * len(probs))()
        for q, kw in zip(arr, probs):
            for k, v in kw.items():
                if isinstance(v, torch.Tensor):
                    v = v.data_ptr()
                elif isinstance(v, _vp):
                    v = v.value
                setattr(q, k, v if v is not None else 0)
        rc = self.lib.msr3d_gemm_multi_f32(len(probs), arr, self.stream)
        _lib.check(rc, "msr3d_gemm_multi_f32")

    @staticmethod
    def _dw(dy, n_out, x, k_in, m_tok, dw, db):
        """dW (n_out, k_in) += dy^T x, db += colsum(dy): reduction over the tokens."""
        return dict(a_kc=0, b_kc=0, M=n_out, N=k_in, K=m_tok, A=dy, lda=n_out, B=x, ldb=k_in, C=dw, ldc=k_in,
                    beta=1.0, colsum=db)

    # ------------------------------------------------------------------ data-only front
    def stage(self, d, anchor_out=None):
        """pad mask, pairwise features, Fourier features, obj_locs copy: one launch, reading the
        batch tensors where they are.  HotPathTrainStep calls it eagerly per batch (outside the
        graph); forward() calls it itself otherwise.  anchor_out = (anchor_locs, anchor_orientation)
        static buffers that receive a copy of the batch's anchor pose in the same launch."""
        e = d["obj_embeds"]
        B, L = e.shape[:2]
        self._ensure(B, L, e.shape[-1], e.device)
        a = self.arena
        dev = e.device
        loc = d["obj_locs"].contiguous()
        valid = d["obj_masks"].contiguous().view(torch.uint8)
        al, ao = d["anchor_locs"].contiguous(), d["anchor_orientation"].contiguous()
        lib = _lib.load()
        if self.anchor:
            # the agent's row in front of every scene's (position + the constant anchor_size, always valid:
            # ose3d_situation.py:336-345) and its orientation as Fourier rows for the orientation encoder (:346-349): the
            # same one launch
            pr = self.pr
            with torch.cuda.device(dev):
                rc = lib.msr3d_scene_prologue_agent(B, L, _ptr(loc), _ptr(valid), _ptr(al), _ptr(ao), _ptr(pr.anchor_size),
                                                    _ptr(self.freqs), 10, ctypes.c_float(1e-10), _ptr(a["pw"]), _ptr(a["ff"]),
                                                    _ptr(a["loc6"]), _ptr(self.pad), _ptr(self.valid),
                                                    _ptr(a["qf"]) if pr.use_orientation else None,
                                                    _ptr(anchor_out[0]) if anchor_out else None,
                                                    _ptr(anchor_out[1]) if anchor_out else None,
                                                    _lib.current_stream_ptr(dev))
            _lib.check(rc, "msr3d_scene_prologue_agent")
            return
        with torch.cuda.device(dev):
            rc = lib.msr3d_scene_prologue(B, L, _ptr(loc), _ptr(valid), _ptr(al), _ptr(ao), _ptr(self.freqs), 10, 1,
                                          ctypes.c_float(1e-10), _ptr(a["pw"]), _ptr(a["ff"]), _ptr(a["loc6"]),
                                          _ptr(self.pad), _ptr(self.valid),
                                          _ptr(anchor_out[0]) if anchor_out else None,
                                          _ptr(anchor_out[1]) if anchor_out else None, _lib.current_stream_ptr(dev))
        _lib.check(rc, "msr3d_scene_prologue")

    # ------------------------------------------------------------------ forward / backward
    def forward(self, embeds):
        self._ran_blocks = self.use_blocks()             # (backward takes the schedule forward took)
        if self._ran_blocks:
            return self.forward_blocks(embeds)
        pr, m, a, dm = self.pr, self.model, self.arena, self.dims
        B, L, M, D, W, H, FF, E, KF, KE, nl = (dm[k] for k in ("B", "L", "M", "D", "W", "H", "FF", "E", "KF", "KE", "nl"))
        dev = embeds.device
        self.lib = lib = _lib.load()
        self.stream = st = _lib.current_stream_ptr(dev)
        train = m.training
        seed = hipops.seed_word(dev)
        anchor = self.anchor
        if anchor:
            a["e61"].view(B, L, KE)[:, 1:].copy_(embeds.reshape(B, L - 1, KE))
            e2 = a["e61"]
        else:
            e2 = embeds.reshape(M, KE)
            e2 = e2 if e2.is_contiguous() else e2.contiguous()
        self.saved_embeds = e2
        layers = list(pr.spatial_encoder)
        self.salts = [[hipops._next_salt() for _ in range(4)] for _ in layers]
        self.ps = []
        for layer in layers:
            sa = layer.self_attn
            self.ps.append((float(sa.dropout.p) if train else 0.0, float(layer.dropout1.p) if train else 0.0,
                            float(layer.dropout2.p) if train else 0.0, float(layer.dropout.p) if train else 0.0))
        same_all = pr.cfg.spatial_encoder.obj_loc_encoding == "same_all"
        with torch.cuda.device(dev):
            rc = lib.msr3d_step_begin(_ptr(a.buf), a.zero_floats, _ptr(seed) if self.bump_seed else None, st)
            _lib.check(rc, "msr3d_step_begin")
            lp = pr.obj_linear_projection
            front = [dict(a_kc=1, b_kc=1, M=M, N=D, K=KE, A=e2, lda=KE, B=lp.weight, ldb=KE, C=a["x0"], ldc=D,
                          bias=lp.bias, beta=1.0)]
            if anchor:
                if pr.use_orientation:
                    oe = pr.orientation_encoder
                    front.append(dict(a_kc=1, b_kc=1, M=B, N=D, K=oe.in_features, A=a["qf"], lda=oe.in_features, B=oe.weight,
                                      ldb=oe.in_features, C=a["a_ori"], ldc=D, bias=oe.bias, beta=1.0))
                self._multi(front)
                ll = pr.loc_layers[0]
                rc = lib.msr3d_anchor_front_fwd(B, L, _ptr(a["x0"]), _ptr(a["a_ori"]), _ptr(pr.anchor_feat),
                                                _ptr(pr.object_type_embedding.weight),
                                                _ptr(pr.object_orientation_feat) if pr.use_orientation else None,
                                                _ptr(a["loc6"]), _ptr(ll[0].weight), _ptr(ll[0].bias), _ptr(ll[1].weight),
                                                _ptr(ll[1].bias), ctypes.c_float(ll[1].eps), _ptr(a["pos"]), _ptr(a["sa"]),
                                                _ptr(a["sta"]), _ptr(a["xin0"]), None, st)
                _lib.check(rc, "msr3d_anchor_front_fwd")
            else:
                self._multi(front)
                le, se = pr.loc_embedding_encoder, pr.size_embedding_encoder
                rc = lib.msr3d_pos_embed_fwd(M, KF, _ptr(a["ff"]), _ptr(a["loc6"]), _ptr(le[0].weight), _ptr(le[0].bias),
                                             _ptr(le[1].weight), _ptr(le[1].bias), ctypes.c_float(le[1].eps),
                                             _ptr(se[0].weight), _ptr(se[0].bias), _ptr(se[1].weight), _ptr(se[1].bias),
                                             ctypes.c_float(se[1].eps), _ptr(a["pos"]), _ptr(a["sa"]), _ptr(a["sta"]),
                                             _ptr(a["sb"]), _ptr(a["stb"]), st)
                _lib.check(rc, "msr3d_pos_embed_fwd")
            for i, layer in enumerate(layers):
                sa = layer.self_attn
                wv, bv = sa._packed[0], sa._packed[1]
                p_attn, p1, p2, p_ffn = self.ps[i]
                s_attn, s_1, s_2, s_ffn = self.salts[i]
                if i == 0 and anchor:       # (msr3d_anchor_front_fwd wrote the layer input)
                    self._strip(M=M, N=W, pro=PRO["plain"], epi=EPI["bias"], b_kc=1, a0=a["xin0"], W=wv, ldw=D, bias=bv,
                                C=a["qkvc0"], ldc=W)
                elif i == 0:
                    self._strip(M=M, N=W, pro=PRO["add"], epi=EPI["bias"], b_kc=1, a0=a["x0"], a1=a["pos"],
                                g1=_ptr(pr.object_type_embedding.weight),        # row 0: every object has type id 0
                                b1=_ptr(pr.object_orientation_feat) if pr.use_orientation else None,
                                o1=a["xin0"], W=wv, ldw=D, bias=bv, C=a["qkvc0"], ldc=W)
                else:
                    prev = layers[i - 1]
                    self._strip(M=M, N=W, pro=PRO["ln"], epi=EPI["bias"], b_kc=1, a0=a[f"ffn{i-1}"], a1=a[f"t{i-1}"],
                                a2=a["pos"] if same_all else None, g1=prev.norm2.weight, b1=prev.norm2.bias,
                                eps1=prev.norm2.eps, p1=self.ps[i - 1][2], salt1=self.salts[i - 1][2], seed=seed,
                                o0=a[f"s3_{i-1}"], ost1=a[f"st3_{i-1}"], o1=a[f"xin{i}"], W=wv, ldw=D, bias=bv,
                                C=a[f"qkvc{i}"], ldc=W)
                q = a[f"qkvc{i}"]
                base, fs = q.data_ptr(), 4
                rc = lib.msr3d_spatial_attn_fwd(B, L, H, D // H, 5, _vp(base), _vp(base + D * fs), _vp(base + 2 * D * fs),
                                                W, _vp(base + 3 * D * fs), W, _ptr(a["pw"]), _ptr(self.pad),
                                                _ptr(a[f"ctx{i}"]), _ptr(a[f"probs{i}"]), hipops.attention_mma(False, training=True), st)
                _lib.check(rc, "msr3d_spatial_attn_fwd")
                self._strip(M=M, N=D, pro=PRO["plain"], epi=EPI["bias"], b_kc=1, a0=a[f"ctx{i}"], W=sa.fc.weight,
                            ldw=D, bias=sa.fc.bias, C=a[f"fc{i}"], ldc=D)
                self._strip(M=M, N=FF, pro=PRO["ln2"], epi=EPI["gelu"], b_kc=1, a0=a[f"fc{i}"], a1=a[f"xin{i}"],
                            g1=sa.layer_norm.weight, b1=sa.layer_norm.bias, eps1=sa.layer_norm.eps, p1=p_attn,
                            salt1=s_attn, g2=layer.norm1.weight, b2=layer.norm1.bias, eps2=layer.norm1.eps, p2=p1,
                            salt2=s_1, seed=seed, o0=a[f"s1_{i}"], ost1=a[f"st1_{i}"], o2=a[f"s2_{i}"],
                            ost2=a[f"st2_{i}"], o1=a[f"t{i}"], W=layer.linear1.weight, ldw=D, bias=layer.linear1.bias,
                            C=a[f"h{i}"], ldc=FF, Cpre=a[f"pre{i}"], p_drop=p_ffn, salt=s_ffn)
                self._multi([dict(a_kc=1, b_kc=1, M=M, N=D, K=FF, A=a[f"h{i}"], lda=FF, B=layer.linear2.weight, ldb=FF,
                                  C=a[f"ffn{i}"], ldc=D, bias=layer.linear2.bias, beta=1.0)])
            last = layers[-1]
            self._strip(M=M, N=E, pro=PRO["ln"], epi=EPI["bias"], b_kc=1, a0=a[f"ffn{nl-1}"], a1=a[f"t{nl-1}"],
                        g1=last.norm2.weight, b1=last.norm2.bias, eps1=last.norm2.eps, p1=self.ps[-1][2],
                        salt1=self.salts[-1][2], seed=seed, o0=a[f"s3_{nl-1}"], ost1=a[f"st3_{nl-1}"], o1=a["tok"],
                        W=m.llm_proj.weight, ldw=D, bias=m.llm_proj.bias, C=a["scene"], ldc=E)
        return a["tok"].view(B, L, D), a["scene"].view(B, L, E)

    def backward(self, g_scene, g_tok):
        if self._ran_blocks:
            return self.backward_blocks(g_scene, g_tok)
        pr, m, a, dm = self.pr, self.model, self.arena, self.dims
        B, L, M, D, W, H, FF, E, KF, KE, nl = (dm[k] for k in ("B", "L", "M", "D", "W", "H", "FF", "E", "KF", "KE", "nl"))
        dev = a.buf.device
        lib = self.lib
        self.stream = st = _lib.current_stream_ptr(dev)
        seed = hipops.seed_word(dev)
        layers = list(pr.spatial_encoder)
        same_all = pr.cfg.spatial_encoder.obj_loc_encoding == "same_all"
        with torch.cuda.device(dev):
            if g_tok is not None:            # a consumer of obj_tokens besides llm_proj
                a["d_tok"].add_(g_tok.reshape(M, D))
            if g_scene is not None:
                g = g_scene.reshape(M, E)
                g = g if g.is_contiguous() else g.contiguous()
                lp = m.llm_proj
                self._multi([dict(a_kc=1, b_kc=0, M=M, N=D, K=E, A=g, lda=E, B=lp.weight, ldb=D, C=a["d_tok"], ldc=D,
                                  beta=1.0),
                             self._dw(g, E, a["tok"], D, M, lp.weight.grad, lp.bias.grad)])
            d_out = a["d_tok"]
            for i in range(nl - 1, -1, -1):
                layer = layers[i]
                sa = layer.self_attn
                wv, bv, gwv, gbv, _dp = sa._packed
                p_attn, p1, p2, p_ffn = self.ps[i]
                s_attn, s_1, s_2, s_ffn = self.salts[i]
                l1, l2 = layer.linear1, layer.linear2
                # LN(norm2)-bwd -> d_ffn; d_h = d_ffn W2; d_pre = dropout-bwd(d_h) * gelu'(pre)
                self._strip(M=M, N=FF, pro=PRO["lnbwd"], epi=EPI["gelubwd"], b_kc=0, a0=d_out, a1=a[f"s3_{i}"],
                            st1=a[f"st3_{i}"], g1=layer.norm2.weight, p1=p2, salt1=s_2, seed=seed, o0=a["d_ffn"],
                            o1=a["d_t"], dg1=layer.norm2.weight.grad, db1=layer.norm2.bias.grad, W=l2.weight, ldw=FF,
                            C=a["d_pre"], ldc=FF, pre_in=a[f"pre{i}"], p_drop=p_ffn, salt=s_ffn)
                self._multi([dict(a_kc=1, b_kc=0, M=M, N=D, K=FF, A=a["d_pre"], lda=FF, B=l1.weight, ldb=D, C=a["d_t"],
                                  ldc=D, beta=1.0),
                             self._dw(a["d_ffn"], D, a[f"h{i}"], FF, M, l2.weight.grad, l2.bias.grad),
                             self._dw(a["d_pre"], FF, a[f"t{i}"], D, M, l1.weight.grad, l1.bias.grad)])
                # LN(norm1), LN(attention tail) bwd -> d_fc, residual gradient; d_ctx = d_fc Wfc
                self._strip(M=M, N=D, pro=PRO["ln2bwd"], epi=EPI["bias"], b_kc=0, a0=a["d_t"], a1=a[f"s1_{i}"],
                            a2=a[f"s2_{i}"], st1=a[f"st1_{i}"], st2=a[f"st2_{i}"], g1=sa.layer_norm.weight,
                            g2=layer.norm1.weight, p1=p_attn, salt1=s_attn, p2=p1, salt2=s_1, seed=seed,
                            o0=a["d_fc"], o1=a[f"d_xin{i}"], dg1=sa.layer_norm.weight.grad, db1=sa.layer_norm.bias.grad,
                            dg2=layer.norm1.weight.grad, db2=layer.norm1.bias.grad, W=sa.fc.weight, ldw=D,
                            C=a["d_ctx"], ldc=D)
                q, gq = a[f"qkvc{i}"], a["d_qkvc"]
                base, gb, fs = q.data_ptr(), gq.data_ptr(), 4
                rc = lib.msr3d_spatial_attn_bwd(B, L, H, D // H, 5, _vp(base), _vp(base + D * fs), _vp(base + 2 * D * fs),
                                                W, _vp(base + 3 * D * fs), W, _ptr(a["pw"]), _ptr(self.pad),
                                                _ptr(a[f"probs{i}"]), _ptr(a["d_ctx"]), _vp(gb), _vp(gb + D * fs),
                                                _vp(gb + 2 * D * fs), W, _vp(gb + 3 * D * fs), W,
                                                hipops.attention_mma(True), st)
                _lib.check(rc, "msr3d_spatial_attn_bwd")
                self._multi([dict(a_kc=1, b_kc=0, M=M, N=D, K=W, A=gq, lda=W, B=wv, ldb=D, C=a[f"d_xin{i}"], ldc=D,
                                  beta=1.0),
                             self._dw(gq, W, a[f"xin{i}"], D, M, gwv, gbv),
                             self._dw(a["d_fc"], D, a[f"ctx{i}"], D, M, sa.fc.weight.grad, sa.fc.bias.grad)])
                d_out = a[f"d_xin{i}"]
            more = same_all and nl > 1
            if self.anchor:
                ll = pr.loc_layers[0]
                tg = pr.object_type_embedding.weight.grad
                lp = pr.obj_linear_projection
                rc = lib.msr3d_anchor_front_bwd(
                    B, L, _ptr(a["d_xin0"]), _ptr(a["d_xin1"]) if more else None,
                    _ptr(a["d_xin2"]) if (more and nl > 2) else None, _ptr(a["sa"]), _ptr(a["sta"]), _ptr(ll[1].weight),
                    _ptr(a["d_la"]), _ptr(ll[1].weight.grad), _ptr(ll[1].bias.grad), _ptr(tg),
                    _ptr(pr.object_orientation_feat.grad) if pr.use_orientation else None, _ptr(lp.bias.grad), _ptr(tg, D),
                    _ptr(pr.anchor_feat.grad), st)
                _lib.check(rc, "msr3d_anchor_front_bwd")
                # loc_layers[0][0] (6 -> 256), the orientation encoder over the agent rows of d xin0 (row stride L * 256),
                # the projection against the features with a zero row per agent (its bias gradient came from the launch above)
                last = [dict(a_kc=0, b_kc=0, M=D, N=6, K=M, A=a["d_la"], lda=D, B=a["loc6"], ldb=6, C=ll[0].weight.grad, ldc=6,
                             beta=1.0, colsum=ll[0].bias.grad),
                        dict(a_kc=0, b_kc=0, M=D, N=KE, K=M, A=a["d_xin0"], lda=D, B=self.saved_embeds, ldb=KE,
                             C=lp.weight.grad, ldc=KE, beta=1.0)]
                if pr.use_orientation:
                    oe = pr.orientation_encoder
                    QF = oe.in_features
                    last.append(dict(a_kc=0, b_kc=0, M=D, N=QF, K=B, A=a["d_xin0"], lda=L * D, B=a["qf"], ldb=QF,
                                     C=oe.weight.grad, ldc=QF, beta=1.0, colsum=oe.bias.grad))
                self._multi(last)
                for p in self._params():
                    self.dp.mark_ready(p)
                return
            le, se = pr.loc_embedding_encoder, pr.size_embedding_encoder
            rc = lib.msr3d_pos_embed_bwd(
                M, _ptr(a["d_xin0"]), _ptr(a["d_xin1"]) if more else None,
                _ptr(a["d_xin2"]) if (more and nl > 2) else None, _ptr(a["sa"]), _ptr(a["sta"]), _ptr(le[1].weight),
                _ptr(a["sb"]), _ptr(a["stb"]), _ptr(se[1].weight), _ptr(a["d_la"]), _ptr(a["d_lb"]),
                _ptr(le[1].weight.grad), _ptr(le[1].bias.grad), _ptr(se[1].weight.grad), _ptr(se[1].bias.grad),
                _ptr(pr.object_type_embedding.weight.grad),
                _ptr(pr.object_orientation_feat.grad) if pr.use_orientation else None, st)
            _lib.check(rc, "msr3d_pos_embed_bwd")
            lp = pr.obj_linear_projection
            last = [self._dw(a["d_la"], D, a["ff"], KF, M, le[0].weight.grad, le[0].bias.grad),
                    dict(a_kc=0, b_kc=0, M=D, N=3, K=M, A=a["d_lb"], lda=D, B=_ptr(a["loc6"], 3), ldb=6,
                         C=se[0].weight.grad, ldc=3, beta=1.0, colsum=se[0].bias.grad),
                    self._dw(a["d_xin0"], D, self.saved_embeds, KE, M, lp.weight.grad, lp.bias.grad)]
            if self.need_d_embeds:     # unfrozen object encoder: d obj_embeds = d_xin0 W_proj, same launch
                last.append(dict(a_kc=1, b_kc=0, M=M, N=KE, K=D, A=a["d_xin0"], lda=D, B=lp.weight, ldb=KE,
                                 C=a["d_emb"], ldc=KE, beta=0.0))
            self._multi(last)
        for p in self._params():
            self.dp.mark_ready(p)


class _PrompterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sched, embeds, *params):
        tok, scene = sched.forward(embeds)
        ctx.sched = sched
        ctx.n = len(params)
        ctx.d_embeds = embeds.requires_grad
        ctx.set_materialize_grads(False)
        return tok, scene

    @staticmethod
    def backward(ctx, g_tok, g_scene):
        sched = ctx.sched
        d_emb = None
        if g_tok is not None or g_scene is not None:
            sched.need_d_embeds = ctx.d_embeds
            sched.backward(g_scene, g_tok)
            if ctx.d_embeds:
                B, L, KE = (sched.dims[k] for k in ("B", "L", "KE"))
                d_emb = sched.arena["d_emb"].view(B, L, KE)
        return (None, d_emb) + (None,) * ctx.n


def attach(model, dp):
    """Give `model` (MSR3DHotPath) its schedule; call after hipops.attach_packed_views."""
    model._schedule = PrompterSchedule(model, dp)
    return model._schedule


def run(model, d):
    """-> d with obj_tokens / scene_embeds / obj_masks / oatt, through the fused schedule."""
    sched = model._schedule
    e = d["obj_embeds"]
    B, L = e.shape[:2]
    sched._ensure(B, L, e.shape[-1], e.device)
    if not d.get("_staged", False):
        sched.stage(d)
    tok, scene = _PrompterFn.apply(sched, e, *sched._params())
    if sched.anchor:
        d["obj_masks"] = sched.valid          # (B, O + 1): the agent's token is always valid
    d["oatt"] = None
    d["obj_tokens"] = tok
    d["scene_embeds"] = scene
    return d
