"""`VISION_REGISTRY["PcdObjEncoder"]` -- per-object point-cloud encoder
(mirror of /root/reference/modules/vision/pcd_pointnet_encoder.py:10-74).

forward(obj_pcds[B,O,P,6], ...) -> (obj_embeds[B,O,768], obj_sem_cls[B,O,607]).
With `freeze=True` (every shipped config) BN runs on its running statistics and the
backbone is evaluated under no_grad.
"""
import torch
from torch import nn

from ..build import VISION_REGISTRY
from ..layers.pointnet import PointNetPP
from ..utils import get_mlp_head


@VISION_REGISTRY.register()
class PcdObjEncoder(nn.Module):
    def __init__(self, cfg, sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None],
                 sa_radii=[0.2, 0.4, None],
                 sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]],
                 dropout=0.1, path=None, freeze=False):
        super().__init__()
        self.freeze = bool(freeze)
        self.pcd_net = PointNetPP(sa_n_points=sa_n_points, sa_n_samples=sa_n_samples,
                                  sa_radii=sa_radii, sa_mlps=sa_mlps)
        # 607 ScanNet classes behind a 384-wide hidden layer, as in the checkpoints (:30)
        self.obj3d_clf_pre_head = get_mlp_head(sa_mlps[-1][-1], 384, 607, dropout=0.3)
        self.dropout = nn.Dropout(dropout)
        if path:   # the shipped yaml carries `path: ""`; only a real checkpoint path is loaded
            state = torch.load(path, map_location="cpu")
            self.load_state_dict(state, strict=False)
        if self.freeze:
            self.requires_grad_(False)

    @staticmethod
    def freeze_bn(m):
        """BatchNorm layers of `m` onto their running statistics."""
        for bn in (l for l in m.modules() if isinstance(l, nn.BatchNorm2d)):
            bn.eval()

    # Opt-in: do not re-encode padding slots.  The dataset pads every scene to 60 objects with one
    # constant cloud (dataset_wrapper.py:156-158) and the reference encodes it again in every slot;
    # with `skip_padded` and `obj_masks` given, masked slots receive the (cached) feature of that
    # cloud instead -- identical values as long as masked slots hold the padding cloud.
    skip_padded = False

    def encode(self, obj_pcds, obj_masks=None, out=None):
        """(B, O, P, C) -> (B, O, D): objects are independent clouds for the backbone."""
        B, O, P, C = obj_pcds.shape
        valid = obj_masks.reshape(B * O) if (self.skip_padded and obj_masks is not None) else None
        o2 = out.view(B * O, -1) if out is not None else None
        return self.pcd_net(obj_pcds.reshape(B * O, P, C), valid=valid, out=o2).reshape(B, O, -1)

    def embed(self, obj_pcds, obj_masks=None, out=None):
        """obj_embeds only: what OSE3DSituation consumes (it takes `[0]` of forward and
        discards the 607-way logits, ose3d_situation.py:285), without the dead head."""
        if not self.freeze:
            return self.encode(obj_pcds, obj_masks, out)
        self.freeze_bn(self.pcd_net)
        with torch.no_grad():
            return self.encode(obj_pcds, obj_masks, out).detach()

    def forward(self, obj_pcds, obj_locs=None, obj_masks=None, obj_sem_masks=None, **kwargs):
        obj_embeds = self.embed(obj_pcds, obj_masks)
        return obj_embeds, self.obj3d_clf_pre_head(obj_embeds)
