"""Generate tests/golden/*.npz by importing the REFERENCE's Python here.

Run in the build container only (needs /root/reference; nothing on the GPU box
reads it):   python tests/golden/make_golden.py

The reference's native ops are CUDA-only, so `pointnet2_utils._ext` is replaced by
the CPU oracle (oracle/pn2.py); everything above it -- QueryAndGroup, SharedMLP,
PointnetSAModule, PointNetPP, PcdObjEncoder, calc_pairwise_locs,
TransformerSpatialEncoderLayer, OSE3DSituation -- is the reference's own code
executed on CPU in fp32.  Import recipe: SURVEY.md §8(c).

Fixtures hold DATA only: inputs, expected outputs, and the weight seed.  Weights
are not stored: `fill_state_dict` regenerates them from (key, shape, seed), the
same function the tests use to fill OUR modules -- which also checks that the
state-dict key sets are identical.
"""
import builtins
import os
import sys
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


# --------------------------------------------------------------------------- shared
def fill_state_dict(sd, seed):
    """Deterministic weights from (key, shape, seed): used for the reference model here
    and for ours in the tests.  BN running_var / LN weights stay positive."""
    import torch
    out = {}
    for k, v in sd.items():
        rng = np.random.default_rng([seed, zlib.crc32(k.encode())])
        if k.endswith("num_batches_tracked"):
            out[k] = torch.tensor(7, dtype=v.dtype)
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy(rng.uniform(0.5, 1.5, v.shape).astype(np.float32))
        elif k.endswith("running_mean"):
            out[k] = torch.from_numpy((0.1 * rng.standard_normal(v.shape)).astype(np.float32))
        elif ".bn.bn.weight" in k or "norm" in k and k.endswith("weight") or \
                k.endswith((".1.weight",)) and v.dim() == 1:
            out[k] = torch.from_numpy(rng.uniform(0.8, 1.2, v.shape).astype(np.float32))
        elif v.dim() >= 2:
            fan_in = int(np.prod(v.shape[1:]))
            out[k] = torch.from_numpy(
                (rng.standard_normal(v.shape) * (1.5 / np.sqrt(fan_in))).astype(np.float32))
        else:
            out[k] = torch.from_numpy((0.05 * rng.standard_normal(v.shape)).astype(np.float32))
        assert tuple(out[k].shape) == tuple(v.shape), k
    return out


def make_scene(seed, B, O, P, n_pad):
    """Small synthetic scenes in the dataset's conventions."""
    from msr3d_amd.synth import synth_batch
    return synth_batch(seed, B, O=O, P=P, n_valid=[O - n_pad, O - n_pad // 2][:B] if B <= 2
                       else None, device="cpu")


# --------------------------------------------------------------------------- import recipe
def import_reference():
    import torch  # noqa: F401
    import transformers  # noqa: F401  (must be first)
    from oracle import pn2

    class Registry:
        def __init__(self, name):
            self.t = {}

        def register(self, obj=None):
            if obj is None:
                def deco(c):
                    self.t[c.__name__] = c
                    return c
                return deco
            self.t[obj.__name__] = obj
            return obj

        def get(self, name):
            return self.t[name]

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "fvcore" not in sys.modules:
        stub("fvcore")
        stub("fvcore.common")
        stub("fvcore.common.registry", Registry=Registry)

    class EasyDict(dict):
        __getattr__ = dict.__getitem__
    if "easydict" not in sys.modules:
        stub("easydict", EasyDict=EasyDict)
    try:
        import timm.models.vision_transformer  # noqa: F401
    except Exception:
        stub("timm")
        stub("timm.models")
        stub("timm.models.vision_transformer", PatchEmbed=object, Attention=object, Mlp=object)
    import transformers.modeling_utils as mu
    for name in ("Conv1D", "find_pruneable_heads_and_indices", "prune_conv1d_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, None)
    try:
        import omegaconf  # noqa: F401
    except Exception:
        oc = types.SimpleNamespace(to_container=lambda c: {k: c[k] for k in c})
        stub("omegaconf", OmegaConf=oc)

    # bare package objects so the reference's __init__ import chains are bypassed
    sys.path.insert(0, REF)
    for pkg, path in [("modules", "modules"), ("modules.layers", "modules/layers"),
                      ("modules.vision", "modules/vision"),
                      ("modules.third_party", "modules/third_party"),
                      ("modules.third_party.pointnet2", "modules/third_party/pointnet2"),
                      ("model", "model"), ("common", "common"), ("optim", "optim")]:
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, path)]
        sys.modules[pkg] = m

    builtins.__POINTNET2_SETUP__ = True
    import importlib
    pu = importlib.import_module("modules.third_party.pointnet2.pointnet2_utils")
    pu._ext = pn2.ext_module()
    sys.modules["pointnet2_utils"] = pu
    sys.modules["pytorch_utils"] = importlib.import_module(
        "modules.third_party.pointnet2.pytorch_utils")
    # ose3d_situation imports optim.utils.no_decay_param_group (unused on the path)
    stub("optim.utils", no_decay_param_group=lambda *a, **k: None)
    importlib.import_module("modules.vision.pcd_pointnet_encoder")
    ose = importlib.import_module("model.ose3d_situation")
    return pu, ose


def ref_cfg(situation_type):
    """Attribute-dict with the values of configs/msr3d.yaml:175-217."""
    from msr3d_amd.config import default_prompter_cfg
    return default_prompter_cfg(situation_type=situation_type, freeze=True)


def capture_encoder_internals(pu, encoder, obj_fts):
    """Run the reference PointNetPP level by level, recording the index ops' outputs."""
    import torch
    rec = {}
    b = obj_fts.shape[0] * obj_fts.shape[1]
    pc = obj_fts.reshape(b, obj_fts.shape[2], obj_fts.shape[3])
    xyz = pc[..., :3].contiguous()
    feats = pc[..., 3:].transpose(1, 2).contiguous()
    with torch.no_grad():
        for lvl, sa in enumerate(encoder.pcd_net.encoder):
            if sa.npoint is not None:
                fidx = pu.furthest_point_sample(xyz, sa.npoint)
                new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), fidx) \
                    .transpose(1, 2).contiguous()
                bidx = pu.ball_query(sa.groupers[0].radius, sa.groupers[0].nsample, xyz, new_xyz)
                rec[f"sa{lvl}_fps_idx"] = fidx.numpy()
                rec[f"sa{lvl}_ball_idx"] = bidx.numpy()
            xyz, feats = sa(xyz, feats)
            rec[f"sa{lvl}_out_first2"] = feats[:2].numpy().copy()
    return rec


def main():
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(4)
    pu, ose = import_reference()

    for variant, situation_type in [("transform", "as_transform_for_objects"),
                                    ("anchor", "as_object")]:
        for seed in (0, 1):
            B, O, P, n_pad = 2, 12, 1024, 5
            batch = make_scene(seed, B, O, P, n_pad)
            cfg = ref_cfg(situation_type)
            model = ose.OSE3DSituation(cfg).eval()
            model.load_state_dict(fill_state_dict(model.state_dict(), seed), strict=True)

            rec = {k: v.numpy() for k, v in batch.items()}
            rec["weight_seed"] = np.int64(seed)
            rec["state_keys"] = np.array(sorted(model.state_dict().keys()))
            rec.update(capture_encoder_internals(pu, model.obj_encoder, batch["obj_fts"]))

            # hooks: per-layer outputs and the fused attention of layer 0
            layer_out, attn0 = [], []
            hs = [l.register_forward_hook(lambda m, i, o: layer_out.append(o[0].detach().numpy()))
                  for l in model.spatial_encoder]
            hs.append(model.spatial_encoder[0].self_attn.register_forward_hook(
                lambda m, i, o: attn0.append(o[1].detach().numpy())))
            dd = {k: v.clone() for k, v in batch.items()}
            with torch.no_grad():
                enc, sem = model.obj_encoder(dd["obj_fts"])
            rec["enc_out"] = enc.numpy()
            rec["sem_cls_first"] = sem[:, :2].numpy()

            # forward + backward of loss = sum(obj_tokens @ llm_proj * fixed_random)
            E = 512
            proj = torch.nn.Linear(256, E)
            proj.load_state_dict(fill_state_dict(proj.state_dict(), seed + 100))
            out = model(dd)
            tokens = out["obj_tokens"]
            scene = proj(tokens)
            g = torch.from_numpy(np.random.default_rng(seed + 7).standard_normal(
                tuple(scene.shape)).astype(np.float32))
            loss = (scene * g).sum()
            loss.backward()
            for h in hs:
                h.remove()

            locs = dd["obj_locs"] if variant == "transform" else None
            rec["obj_tokens"] = tokens.detach().numpy()
            rec["obj_masks_out"] = out["obj_masks"].numpy()
            rec["scene_embeds"] = scene.detach().numpy()
            rec["loss_grad_seed"] = np.int64(seed + 7)
            rec["loss"] = np.float64(loss.item())
            for i, lo in enumerate(layer_out):
                rec[f"layer{i}_out"] = lo
            rec["layer0_fused_attn"] = attn0[0]
            if locs is not None:
                from modules.utils import calc_pairwise_locs
                rec["pairwise_locs"] = calc_pairwise_locs(
                    locs[:, :, :3], locs[:, :, 3:], pairwise_rel_type="center",
                    spatial_dist_norm=True, spatial_dim=5).numpy()
            # gradients: norm + sum + first entries per trainable tensor (full grads = 17 MB)
            names, norms, sums, heads = [], [], [], []
            for n, p in list(model.named_parameters()) + [("llm_proj." + k, v) for k, v in
                                                          proj.named_parameters()]:
                if p.grad is None:
                    continue
                gflat = p.grad.detach().double().flatten()
                names.append(n)
                norms.append(gflat.norm().item())
                sums.append(gflat.sum().item())
                heads.append(np.pad(gflat[:8].numpy(), (0, max(0, 8 - gflat.numel()))))
            rec["grad_names"] = np.array(names)
            rec["grad_norms"] = np.array(norms)
            rec["grad_sums"] = np.array(sums)
            rec["grad_heads"] = np.stack(heads)

            path = os.path.join(HERE, f"prompter_{variant}_seed{seed}.npz")
            np.savez_compressed(path, **rec)
            print("wrote", path, os.path.getsize(path) // 1024, "KiB", "loss", loss.item())


if __name__ == "__main__":
    main()
