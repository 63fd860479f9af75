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
        self.pcd_net = PointNetPP(sa_n_points=sa_n_points, sa_n_samples=sa_n_samples,
                                  sa_radii=sa_radii, sa_mlps=sa_mlps)
        self.obj3d_clf_pre_head = get_mlp_head(sa_mlps[-1][-1], 384, 607, dropout=0.3)
        self.dropout = nn.Dropout(dropout)
        if path:   # the shipped yaml carries `path: ""`; only a real path is loaded
            self.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
        self.freeze = freeze
        if freeze:
            for p in self.parameters():
                p.requires_grad = False

    def freeze_bn(self, m):
        for layer in m.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()

    def encode(self, obj_pcds):
        B, O = obj_pcds.shape[:2]
        flat = obj_pcds.reshape(B * O, obj_pcds.size(2), obj_pcds.size(3))
        return self.pcd_net(flat).reshape(B, O, -1)

    def embed(self, obj_pcds):
        """obj_embeds only: what OSE3DSituation consumes (it takes `[0]` of forward and
        discards the 607-way logits, ose3d_situation.py:285), without the dead head."""
        if self.freeze:
            self.freeze_bn(self.pcd_net)
            with torch.no_grad():
                return self.encode(obj_pcds).detach()
        return self.encode(obj_pcds)

    def forward(self, obj_pcds, obj_locs=None, obj_masks=None, obj_sem_masks=None, **kwargs):
        obj_embeds = self.embed(obj_pcds)
        obj_sem_cls = self.obj3d_clf_pre_head(obj_embeds)
        return obj_embeds, obj_sem_cls
