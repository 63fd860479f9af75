"""Minimal attribute-dict config (stands in for OmegaConf/EasyDict nodes) and the
model block of the reference's shipped configs.

`default_prompter_cfg()` restates /root/reference/configs/msr3d.yaml:175-217 (the
`model.prompter` block, identical in all five shipped train configs -- SURVEY.md §5).
"""
import copy


class AttrDict(dict):
    """dict with attribute access, recursive on construction; `.get` as in OmegaConf."""

    def __init__(self, *a, **kw):
        super().__init__()
        for k, v in dict(*a, **kw).items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return [cls._wrap(x) for x in v]
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = self._wrap(v)

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def cfg2dict(cfg):
    """common/type_utils.py:6-7 -- config node -> plain python containers."""
    try:
        from omegaconf import DictConfig, ListConfig, OmegaConf
        if isinstance(cfg, (DictConfig, ListConfig)):
            return OmegaConf.to_container(cfg)
    except ImportError:
        pass
    if isinstance(cfg, dict):
        return {k: cfg2dict(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [cfg2dict(v) for v in cfg]
    return cfg


def default_prompter_cfg(situation_type="as_transform_for_objects", freeze=True, hidden_size=256,
                         num_layers=3, dropout=0.1):
    return AttrDict({
        "model": {
            "name": "OSE3DSituation",
            "situation_type": situation_type,
            "scene_token_len": 60,
            "loc_fourier_dim": 63,
            "hidden_size": hidden_size,
            "label_size": 300,
            "vision_backbone_name": "gtpcd",
            "use_spatial_attn": True,
            "use_anchor": True,
            "use_orientation": True,
            "fourier_size": 84,
            "attn_flat": {"use_attn_flat": False, "mcan_flat_mlp_size": 512,
                          "mcan_flat_glimpses": 1, "mcan_flat_out_size": 1024},
            "vision": {
                "name": "PcdObjEncoder",
                "args": {
                    "sa_n_points": [32, 16, None],
                    "sa_n_samples": [32, 32, None],
                    "sa_radii": [0.2, 0.4, None],
                    "sa_mlps": [[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]],
                    "dropout": 0.1,
                    "freeze": freeze,
                    "path": None,
                },
            },
            "spatial_encoder": {
                "dim_loc": 6, "num_attention_heads": 8, "dim_feedforward": 2048,
                "dropout": dropout, "activation": "gelu", "spatial_multihead": True,
                "spatial_dim": 5, "spatial_dist_norm": True, "spatial_attn_fusion": "cond",
                "num_layers": num_layers, "obj_loc_encoding": "same_all",
                "pairwise_rel_type": "center",
            },
        }
    })
