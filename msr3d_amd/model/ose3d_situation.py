"""`MODEL_REGISTRY["OSE3DSituation"]` -- the situated object-centric scene encoder.

Mirror of /root/reference/model/ose3d_situation.py:157-454 for the configurations
the shipped yaml files select (vision_backbone_name 'gtpcd', PcdObjEncoder,
spatial attention, no attn_flat; situation_type in {'as_object', 'as_object_add_loc',
'as_embedding', 'as_transform_for_objects'}).  Constructor `(cfg)` reading
`cfg.model.*`; `forward(data_dict) -> data_dict` adding `obj_tokens [B,L,H]`,
`obj_masks [B,L] bool (True = valid)`, `oatt`; attributes `.obj_encoder`, `.device`;
parameter names identical so reference checkpoints load with `strict=True`.

Out of scope here: the 'gt' semantic backbone,
PointBERT encoder, attn_flat pooling, cross-attention / DiT situation layers --
constructing them raises NotImplementedError naming the row.
"""
import math

import torch
import torch.nn as nn

from .. import hipops
from ..modules.build import build_module
from ..modules.layers.transformers import (TransformerEncoderLayer,
                                           TransformerSpatialEncoderLayer)
from ..modules.utils import (calc_pairwise_locs, layer_repeat, maybe_autocast,
                             transform_to_agent_coor)
from ..modules.weights import _init_weights_bert
from .build import MODEL_REGISTRY, BaseModel

_ANCHOR_TOKEN_TYPES = ("as_object", "as_object_add_loc")
_LOC_EMBED_TYPES = ("as_object_add_loc", "as_embedding", "as_transform_for_objects")
_UNSUPPORTED_TYPES = ("as_cross_attention", "as_dit_attention")


def generate_fourier_features(pos, num_bands=10, max_freq=15, concat_pos=True, sine_only=False):
    """(B,N,C) -> (B,N,C*(2*num_bands)+C): [pos, sin(pi*pos*f), cos(pi*pos*f)] with
    f = linspace(1, max_freq, num_bands); inside each block the layout is (coordinate,
    band) flattened (ose3d_situation.py:31-59)."""
    B = pos.shape[0]
    freqs = torch.linspace(1.0, max_freq, steps=num_bands, device=pos.device)
    scaled = (pos.unsqueeze(-1) * freqs).reshape(B, -1, pos.shape[2] * num_bands)
    if sine_only:
        feats = torch.sin(math.pi * scaled)
    else:
        feats = torch.cat([torch.sin(math.pi * scaled), torch.cos(math.pi * scaled)], dim=-1)
    return torch.cat([pos, feats], dim=-1) if concat_pos else feats


@MODEL_REGISTRY.register()
class OSE3DSituation(BaseModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.cfg = c = cfg.model
        H = c.hidden_size
        self.vision_backbone_name = c.vision_backbone_name
        self.use_spatial_attn = c.use_spatial_attn
        self.use_anchor = c.use_anchor
        self.use_orientation = c.use_orientation

        if self.use_anchor:
            self.anchor_feat = nn.Parameter(torch.zeros(1, 1, H))
            self.anchor_size = nn.Parameter(torch.ones(1, 1, 3), requires_grad=False)
        if self.use_orientation:
            self.object_orientation_feat = nn.Parameter(torch.zeros(1, 1, H))
            self.orientation_encoder = nn.Linear(c.fourier_size, H)
        self.object_type_embedding = nn.Embedding(2, embedding_dim=H)

        if self.vision_backbone_name == "gt":
            raise NotImplementedError("'gt' semantic backbone is outside the hot path (the "
                                      "reference asserts False on it, ose3d_situation.py:301-302)")
        self.obj_encoder = build_module("vision", c.vision)
        if c.vision.name != "PcdObjEncoder":
            raise NotImplementedError(f"vision encoder {c.vision.name} is outside the hot path")
        self.obj_linear_projection = nn.Linear(c.vision.args.sa_mlps[-1][-1], H)

        se = c.spatial_encoder
        if c.use_spatial_attn:
            layer = TransformerSpatialEncoderLayer(
                H, nhead=se.num_attention_heads, dim_feedforward=se.dim_feedforward,
                dropout=se.dropout, activation=se.activation, spatial_dim=se.spatial_dim,
                spatial_multihead=se.spatial_multihead,
                spatial_attn_fusion=se.spatial_attn_fusion)
        else:
            layer = TransformerEncoderLayer(
                H, nhead=se.num_attention_heads, dim_feedforward=se.dim_feedforward,
                dropout=se.dropout, activation=se.activation)
        self.spatial_encoder = layer_repeat(layer, se.num_layers)

        if se.obj_loc_encoding in ("same_0", "same_all"):
            n_loc = 1
        elif se.obj_loc_encoding == "diff_all":
            n_loc = se.num_layers
        else:
            raise ValueError(f"obj_loc_encoding {se.obj_loc_encoding}")
        self.loc_layers = layer_repeat(nn.Sequential(nn.Linear(se.dim_loc, H), nn.LayerNorm(H)),
                                       n_loc)
        self.spatial_encoder.apply(_init_weights_bert)
        self.loc_layers.apply(_init_weights_bert)

        if c.attn_flat.use_attn_flat:
            raise NotImplementedError("attn_flat pooling is outside the hot path "
                                      "(no shipped config enables it)")
        if self.use_anchor:
            nn.init.normal_(self.anchor_feat, std=0.02)

        self.situation_type = c.get("situation_type", "as_object")
        if self.situation_type in _UNSUPPORTED_TYPES:
            raise NotImplementedError(f"situation_type {self.situation_type} is outside the hot "
                                      "path (never configured; SURVEY.md §2 row 15)")
        if self.situation_type in _LOC_EMBED_TYPES:
            self.loc_embedding_encoder = nn.Sequential(nn.Linear(c.loc_fourier_dim, H),
                                                       nn.LayerNorm(H))
            self.size_embedding_encoder = nn.Sequential(nn.Linear(3, H), nn.LayerNorm(H))

    @property
    def device(self):
        return next(self.parameters()).device

    # ------------------------------------------------------------------ pieces
    def encode_objects(self, obj_fts, obj_masks=None, out=None):
        """Frozen/eval backbone features (B,O,768).  Uses the encoder's `embed` (no dead
        classification head) when it has one, else `forward(...)[0]` like the reference.
        obj_masks only matters to an encoder with `skip_padded` set."""
        enc = self.obj_encoder
        if hasattr(enc, "embed"):
            masks = obj_masks if getattr(enc, "skip_padded", False) else None
            return enc.embed(obj_fts, masks, out) if out is not None else enc.embed(obj_fts, masks)
        res = enc(obj_fts)[0]
        return out.copy_(res) if out is not None else res

    def forward_gtpcd(self, data_dict):
        embeds = data_dict.get("obj_embeds")          # precomputed by a split (graphed) step
        if embeds is None:
            embeds = self.encode_objects(data_dict["obj_fts"], data_dict.get("obj_masks"))
        return hipops.module_linear(self.obj_linear_projection, embeds)

    def _with_anchor_token(self, data_dict, feat, mask, loc, type_emb, ori_feat):
        """Prepend the agent ("self") token: learnt feature, fourier-encoded orientation,
        location = anchor position + learnt-constant size, type id 1."""
        B = feat.size(0)
        dev = feat.device
        if data_dict["anchor_locs"].shape[-1] != 3:
            raise AssertionError("anchor_locs must be (B, 3)")
        a_feat = self.anchor_feat.expand(B, -1, -1)
        a_mask = torch.zeros((B, 1), device=dev, dtype=torch.bool)
        a_loc = torch.cat((data_dict["anchor_locs"].unsqueeze(1),
                           self.anchor_size.expand(B, -1, -1)), dim=-1)
        a_type = self.object_type_embedding(torch.ones((B, 1), dtype=torch.long, device=dev))
        feat = torch.cat((a_feat, feat), dim=1)
        mask = torch.cat((a_mask, mask), dim=1)
        loc = torch.cat((a_loc, loc), dim=1)
        type_emb = torch.cat((a_type, type_emb), dim=1)
        if self.use_orientation:
            a_ori = hipops.module_linear(
                self.orientation_encoder,
                generate_fourier_features(data_dict["anchor_orientation"].unsqueeze(1)))
            ori_feat = torch.cat((a_ori, ori_feat), dim=1)
        return feat, mask, loc, type_emb, ori_feat

    @staticmethod
    def _lin_ln(seq, x):
        """nn.Sequential(Linear, LayerNorm) with the Linear on the HIP GEMM."""
        return hipops.dropout_add_layernorm(hipops.module_linear(seq[0], x), None, seq[1])

    def _query_pos(self, layer_idx, loc, data_dict):
        """Positional term added to the tokens before a layer."""
        se = self.cfg.spatial_encoder
        if se.obj_loc_encoding == "diff_all":
            return self._lin_ln(self.loc_layers[layer_idx], loc)
        st = self.situation_type
        centre, size = loc[:, :, :3], loc[:, :, 3:]
        loc_enc, size_enc = (getattr(self, "loc_embedding_encoder", None),
                             getattr(self, "size_embedding_encoder", None))
        if st == "as_object_add_loc":
            return (self._lin_ln(loc_enc, generate_fourier_features(centre))
                    + self._lin_ln(size_enc, size))
        if st == "as_embedding":
            L = loc.size(1)
            sit_loc = data_dict["anchor_locs"].unsqueeze(1).expand(-1, L, -1)
            sit_ori = data_dict["anchor_orientation"].unsqueeze(1).expand(-1, L, -1)
            return (self._lin_ln(loc_enc, generate_fourier_features(centre))
                    + self._lin_ln(size_enc, size)
                    + hipops.module_linear(self.orientation_encoder,
                                           generate_fourier_features(sit_ori))
                    + self._lin_ln(loc_enc, generate_fourier_features(sit_loc)))
        if st == "as_transform_for_objects":
            if loc.is_cuda and loc.dtype == torch.float32 and not loc.requires_grad:
                # one HIP launch: agent-frame transform + Fourier features (data only, no grad)
                feats = hipops.agent_fourier(loc, data_dict["anchor_locs"],
                                             data_dict["anchor_orientation"])
            else:
                agent = transform_to_agent_coor(centre, data_dict["anchor_locs"],
                                                data_dict["anchor_orientation"])
                feats = generate_fourier_features(agent)
            return self._lin_ln(loc_enc, feats) + self._lin_ln(size_enc, size)
        return self._lin_ln(self.loc_layers[0], loc)

    # ------------------------------------------------------------------ forward
    def forward(self, data_dict):
        """data_dict keys: obj_fts (B,N,P,6) f32, obj_masks (B,N) bool True=real object,
        obj_locs (B,N,6) [centre, size], anchor_locs (B,3), anchor_orientation (B,4 xyzw);
        optional single_obj (encoder-only shortcut, ose3d_situation.py:307-313)."""
        if "single_obj" in data_dict:
            data_dict["single_obj_token"] = self.forward_gtpcd({"obj_fts": data_dict["single_obj"]})
            return data_dict

        feat = self.forward_gtpcd(data_dict)
        mask = ~data_dict["obj_masks"]                     # True = padded
        B, N = feat.shape[:2]
        dev = feat.device
        loc = data_dict["obj_locs"]
        valid_out = data_dict["obj_masks"]                 # returned as is unless a token is prepended
        # every object has type id 0: row 0 of the table, broadcast (an nn.Embedding lookup of a
        # constant would cost a gather forward and a 90 us scatter kernel backward)
        if self.use_anchor and self.situation_type in _ANCHOR_TOKEN_TYPES:
            type_emb = self.object_type_embedding.weight[0].expand(B, N, -1)
            ori_feat = self.object_orientation_feat.expand(B, N, -1) if self.use_orientation else None
            feat, mask, loc, type_emb, ori_feat = self._with_anchor_token(
                data_dict, feat, mask, loc, type_emb, ori_feat)
            feat = feat + ori_feat + type_emb if self.use_orientation else feat + type_emb
            valid_out = ~mask
        else:
            # same sum, one launch each way: the two constants go on in one kernel, and their
            # gradients (column sums of d feat) straight onto the parameters' gradients
            feat = hipops.add_token_constants(
                feat, self.object_type_embedding.weight, 0,
                self.object_orientation_feat if self.use_orientation else None)

        se = self.cfg.spatial_encoder
        pairwise_locs = None
        if self.cfg.use_spatial_attn:
            if (loc.is_cuda and loc.dtype == torch.float32 and not loc.requires_grad
                    and se.pairwise_rel_type == "center" and se.spatial_dist_norm
                    and se.spatial_dim == 5 and loc.size(1) <= 128):
                pairwise_locs = hipops.pairwise_locs_center5(loc)      # one HIP launch
            else:
                pairwise_locs = calc_pairwise_locs(
                    loc[:, :, :3], loc[:, :, 3:], pairwise_rel_type=se.pairwise_rel_type,
                    spatial_dist_norm=se.spatial_dist_norm, spatial_dim=se.spatial_dim)

        x = feat
        with maybe_autocast(self, enabled=False):          # the encoder always runs in fp32
            # 'same_*': the positional term does not depend on the layer (same inputs, shared
            # weights) -- the reference recomputes it three times (ose3d_situation.py:380-410),
            # here it is computed once and autograd sums the three uses.
            shared_pos = None
            if se.obj_loc_encoding != "diff_all":
                shared_pos = self._query_pos(0, loc, data_dict)
            for i, layer in enumerate(self.spatial_encoder):
                if se.obj_loc_encoding == "diff_all":
                    x = x + self._query_pos(i, loc, data_dict)
                elif se.obj_loc_encoding == "same_all" or i == 0:
                    x = x + shared_pos
                if self.cfg.use_spatial_attn:
                    x, _ = layer(x, pairwise_locs, tgt_key_padding_mask=mask)
                else:
                    x, _ = layer(x, tgt_key_padding_mask=mask)

        data_dict["oatt"] = None
        data_dict["obj_tokens"] = x
        data_dict["obj_masks"] = valid_out
        return data_dict
