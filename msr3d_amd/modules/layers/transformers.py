"""Spatial self-attention layers of the situated scene encoder.

Mirror of /root/reference/modules/layers/transformers.py:167-252
(MultiHeadAttentionSpatial) and :298-329 (TransformerSpatialEncoderLayer): same
constructor arguments, forward signatures, return values and parameter names
(`w_qs, w_ks, w_vs, fc, layer_norm, lang_cond_fc | pairwise_loc_fc`, `linear1,
linear2, norm1, norm2`).  This file is the composite (torch-op) formulation, fp32;
the fused MFMA kernels are selected by OSE3DSituation when available.
"""
import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from ... import fused_layer, hipops
from ..utils import get_activation_fn


class MultiHeadAttentionSpatial(nn.Module):
    def __init__(self, d_model, n_head, dropout=0.1, spatial_multihead=True, spatial_dim=5,
                 spatial_attn_fusion="mul"):
        super().__init__()
        assert d_model % n_head == 0, "d_model: %d, n_head: %d" % (d_model, n_head)
        self.n_head = n_head
        self.d_model = d_model
        self.d_per_head = d_model // n_head
        self.spatial_multihead = spatial_multihead
        self.spatial_dim = spatial_dim
        self.spatial_attn_fusion = spatial_attn_fusion

        self.w_qs = nn.Linear(d_model, d_model)
        self.w_ks = nn.Linear(d_model, d_model)
        self.w_vs = nn.Linear(d_model, d_model)
        self.fc = nn.Linear(d_model, d_model)
        self.dropout = nn.Dropout(p=dropout)
        self.layer_norm = nn.LayerNorm(d_model)

        self.spatial_n_head = n_head if spatial_multihead else 1
        if spatial_attn_fusion in ("mul", "bias", "add"):
            self.pairwise_loc_fc = nn.Linear(spatial_dim, self.spatial_n_head)
        elif spatial_attn_fusion == "ctx":
            self.pairwise_loc_fc = nn.Linear(spatial_dim, d_model)
        elif spatial_attn_fusion == "cond":
            self.lang_cond_fc = nn.Linear(d_model, self.spatial_n_head * (spatial_dim + 1))
        else:
            raise NotImplementedError("unsupported spatial_attn_fusion %s" % spatial_attn_fusion)

    # ---- packed q|k|v|cond projection (self-attention with 'cond' fusion) --------------
    def pack_groups(self):
        """Weights, then biases, that should be contiguous in the flat parameter/gradient
        buffers so the four projections run as ONE GEMM on zero-copy views."""
        if self.spatial_attn_fusion != "cond":
            return []
        mods = (self.w_qs, self.w_ks, self.w_vs, self.lang_cond_fc)
        return [[m.weight for m in mods], [m.bias for m in mods]]

    def set_packed(self, packed, members):
        self._packed, self._packed_members = packed, members

    def _heads(self, x):
        """(B, T, H*dh) -> (H, B, T, dh)"""
        B, T, _ = x.shape
        return x.view(B, T, self.n_head, self.d_per_head).permute(2, 0, 1, 3)

    def _loc_term(self, residual, q, pairwise_locs):
        """Spatial term per (head, batch, query, key)."""
        fusion = self.spatial_attn_fusion
        if fusion in ("mul", "bias", "add"):
            loc = hipops.module_linear(self.pairwise_loc_fc, pairwise_locs).permute(3, 0, 1, 2)
            if fusion == "mul":
                loc = F.relu(loc)
            if not self.spatial_multihead:
                loc = loc.expand(self.n_head, -1, -1, -1)
            return loc
        if fusion == "ctx":
            B, L, T, _ = pairwise_locs.shape
            loc = hipops.module_linear(self.pairwise_loc_fc, pairwise_locs).view(
                B, L, T, self.n_head, self.d_per_head)
            return torch.einsum("hblk,blthk->hblt", q, loc) / math.sqrt(self.d_per_head)
        # 'cond': per-token weights over the spatial features + a per-token bias, then sigmoid
        B, L, _ = residual.shape
        w = hipops.module_linear(self.lang_cond_fc, residual).view(
            B, L, self.spatial_n_head, self.spatial_dim + 1)
        w = w.permute(2, 0, 1, 3)                                                 # (h?, B, L, 1+S)
        if self.spatial_n_head == 1:
            w = w.expand(self.n_head, -1, -1, -1)
        bias, w = w[..., :1], w[..., 1:]
        return torch.sigmoid(torch.einsum("hbld,bltd->hblt", w, pairwise_locs) + bias)

    def forward(self, q, k, v, pairwise_locs, key_padding_mask=None, txt_embeds=None):
        residual = q
        if (self.spatial_attn_fusion == "cond" and getattr(self, "use_fused_core", True)
                and hipops.spatial_attn_cond_supported(
                    q, self.n_head, self.spatial_dim, self.spatial_n_head)):
            # ONE projection GEMM for [q | k | v | cond] (self-attention: q is k is v), then
            # the fused HIP core: scores + spatial term + mask + softmax + PV in one launch
            packed = getattr(self, "_packed", None)
            if q is k and k is v and packed is not None and torch.is_grad_enabled():
                qkvc = hipops.linear_packed(q, packed, self._packed_members)
            elif q is k and k is v:
                w = torch.cat([self.w_qs.weight, self.w_ks.weight, self.w_vs.weight,
                               self.lang_cond_fc.weight], 0)
                bias = torch.cat([self.w_qs.bias, self.w_ks.bias, self.w_vs.bias,
                                  self.lang_cond_fc.bias], 0)
                qkvc = hipops.linear(q, w, bias)
            else:
                qkvc = torch.cat([hipops.module_linear(self.w_qs, q), hipops.module_linear(self.w_ks, k),
                                  hipops.module_linear(self.w_vs, v),
                                  hipops.module_linear(self.lang_cond_fc, residual)], -1)
            ctx, probs = hipops.spatial_attn_cond(qkvc, pairwise_locs, key_padding_mask, self.n_head,
                                                  self.d_model)
            out = hipops.dropout_add_layernorm(hipops.module_linear(self.fc, ctx), residual,
                                               self.layer_norm, self.dropout.p, self.training)
            return out, probs.permute(1, 0, 2, 3)
        qh = self._heads(hipops.module_linear(self.w_qs, q))
        kh = self._heads(hipops.module_linear(self.w_ks, k))
        vh = self._heads(hipops.module_linear(self.w_vs, v))
        attn = torch.einsum("hblk,hbtk->hblt", qh, kh) / math.sqrt(self.d_per_head)
        loc = self._loc_term(residual, qh, pairwise_locs)
        multiplicative = self.spatial_attn_fusion in ("mul", "cond")

        if key_padding_mask is not None:
            mask = key_padding_mask[None, :, None, :]                             # True = padded key
            attn = attn.masked_fill(mask, float("-inf"))
            loc = loc.masked_fill(mask, 0.0 if multiplicative else float("-inf"))

        if self.spatial_attn_fusion == "add":
            fused = (torch.softmax(attn, 3) + torch.softmax(loc, 3)) / 2
        else:
            logits = torch.log(torch.clamp(loc, min=1e-6)) + attn if multiplicative else loc + attn
            fused = torch.softmax(logits, 3)
        # (the reference asserts "no NaN" here, a host sync per layer: transformers.py:246.  A
        # fully padded sample is the only way to produce one; callers guarantee >= 1 valid key.)

        out = torch.einsum("hblt,hbtv->hblv", fused, vh)
        B, L = q.shape[:2]
        out = out.permute(1, 2, 0, 3).reshape(B, L, self.d_model)
        out = hipops.dropout_add_layernorm(hipops.module_linear(self.fc, out), residual,
                                           self.layer_norm, self.dropout.p, self.training)
        return out, fused


class TransformerEncoderLayer(nn.Module):
    """Post-/pre-norm encoder layer over nn.MultiheadAttention (transformers.py:120-164);
    kept because TransformerSpatialEncoderLayer derives from it and `use_spatial_attn: False`
    selects it."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, batch_first=True, dropout=0.1,
                 activation="relu", prenorm=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout,
                                               batch_first=batch_first)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = get_activation_fn(activation)
        self.prenorm = prenorm

    def _ffn(self, x):
        if self.activation is F.gelu:     # GELU rides in the first GEMM's epilogue
            h = hipops.module_linear(self.linear1, x, gelu=True)
        else:
            h = self.activation(hipops.module_linear(self.linear1, x))
        return hipops.module_linear(self.linear2, self.dropout(h))

    def forward(self, tgt, tgt_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None):
        x = self.norm1(tgt) if self.prenorm else tgt
        a, attn_w = self.self_attn(query=x, key=x, value=x, attn_mask=tgt_mask,
                                   key_padding_mask=tgt_key_padding_mask)
        tgt = tgt + self.dropout1(a)
        tgt = self.norm2(tgt) if self.prenorm else self.norm1(tgt)
        tgt = tgt + self.dropout2(self._ffn(tgt))
        if not self.prenorm:
            tgt = self.norm2(tgt)
        return tgt, attn_w


class TransformerSpatialEncoderLayer(TransformerEncoderLayer):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 spatial_multihead=True, spatial_dim=5, spatial_attn_fusion="mul"):
        super().__init__(d_model, nhead, dim_feedforward=dim_feedforward, dropout=dropout,
                         activation=activation)
        del self.self_attn
        self.self_attn = MultiHeadAttentionSpatial(
            d_model, nhead, dropout=dropout, spatial_multihead=spatial_multihead,
            spatial_dim=spatial_dim, spatial_attn_fusion=spatial_attn_fusion)

    def forward(self, tgt, tgt_pairwise_locs, tgt_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None):
        # NB the attention block already returns LN(out + x); the layer then adds x AGAIN and
        # normalises (transformers.py:251 then :324-325) -- a double residual, kept as is.
        if fused_layer.eligible(self, tgt, tgt_pairwise_locs):
            # training on the flat-buffer engine: the whole layer as one hand-scheduled node
            return fused_layer.spatial_layer(self, tgt, tgt_pairwise_locs, tgt_key_padding_mask)
        a, attn_w = self.self_attn(tgt, tgt, tgt, tgt_pairwise_locs,
                                   key_padding_mask=tgt_key_padding_mask)
        tgt = hipops.dropout_add_layernorm(a, tgt, self.norm1, self.dropout1.p, self.training)
        tgt = hipops.dropout_add_layernorm(self._ffn(tgt), tgt, self.norm2, self.dropout2.p,
                                           self.training)
        return tgt, attn_w
