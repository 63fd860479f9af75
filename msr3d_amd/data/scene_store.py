"""HBM-resident scan arena.

The reference keeps every visited scan in a host-side global cache as a dictionary
{instance id: float64 (n_i, 6) array}, built with one boolean mask over all points per
instance (/root/reference/data/datasets/scannet_base.py:57-67,
data/datasets/scan_data_loader.py:83-94,191-192; cache: data/datasets/msr3d.py:163-179) and
ships 1.47 MB per sample to the GPU every step.  An MI355X has 288 GB: the whole of ScanNet
(~1500 scans x ~150 k points x 15 B) is ~3.4 GB, so the scans live on the device, each stored
once in instance-sorted order -- xyz f32 and rgb u8 exactly as on disk -- and a sample is
described by (row offset, row count) pairs.
"""
import ctypes

import numpy as np
import torch

from .. import _lib

SEG_CHUNK = 256          # MSR3D_SEG_CHUNK
SEG_MAX_SLOTS = 8192     # MSR3D_SEG_MAX_SLOTS


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def load_scan_pth(path):
    """`scan_data/pcd_with_global_alignment/<scan>.pth`: a tuple whose first, second and LAST
    members are points (N,3), colors (N,3) in 0..255 and instance_labels (N,)
    (scannet_base.py:57-59; 3RScan / ARKit files use member 2, scan_data_loader.py:84-85,135-136,
    which is also the last)."""
    pcd_data = torch.load(path, weights_only=False)
    return np.asarray(pcd_data[0]), np.asarray(pcd_data[1]), np.asarray(pcd_data[-1])


class SceneStore:
    """Append-only arena of instance-sorted scans on one device."""

    def __init__(self, device="cuda", capacity_points=1 << 20):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SceneStore lives in GPU memory; there is no host fallback")
        self.points = torch.empty((capacity_points, 3), dtype=torch.float32, device=self.device)
        self.colors = torch.empty((capacity_points, 3), dtype=torch.uint8, device=self.device)
        self.tail = 0
        self.scans = {}          # scan_id -> dict(base, inst_ids, offsets (np.int64, len n+1), slot)

    # ------------------------------------------------------------------
    def _reserve(self, rows):
        cap = self.points.shape[0]
        if self.tail + rows <= cap:
            return
        new_cap = max(cap * 2, self.tail + rows)
        for name in ("points", "colors"):
            old = getattr(self, name)
            new = torch.empty((new_cap, 3), dtype=old.dtype, device=self.device)
            new[:self.tail].copy_(old[:self.tail])
            setattr(self, name, new)

    def add_scan(self, scan_id, points, colors, instance_labels, inst_ids=None):
        """points (N,3) float, colors (N,3) 0..255, instance_labels (N,) integer (negative =
        unlabelled).  inst_ids: the instance ids that become objects -- default
        range(instance_labels.max()+1) like scannet_base.py:65 / scan_data_loader.py:191-192; pass
        the keys of inst_to_label for 3RScan / ARKit (scan_data_loader.py:89-93,140-148).
        Instances without points are dropped (the reference would fail on them in
        np.random.choice).  Returns the list of object ids kept."""
        lib = _lib.load()
        pts = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float32)).to(self.device)
        col_np = np.asarray(colors)
        if col_np.dtype != np.uint8:
            if (col_np != np.round(col_np)).any() or col_np.min() < 0 or col_np.max() > 255:
                raise ValueError("colors must hold integers 0..255 (the on-disk format)")
            col_np = col_np.astype(np.uint8)
        col = torch.as_tensor(np.ascontiguousarray(col_np)).to(self.device)
        lab_np = np.ascontiguousarray(instance_labels, dtype=np.int64)
        lab = torch.as_tensor(lab_np).to(self.device)
        n = int(pts.shape[0])
        if col.shape != (n, 3) or lab.shape != (n,) or pts.shape != (n, 3):
            raise ValueError("points (N,3), colors (N,3), instance_labels (N,) expected")
        if inst_ids is None:
            inst_ids = range(int(lab_np.max()) + 1) if n else []
        inst_ids = [int(i) for i in inst_ids]
        if len(inst_ids) > SEG_MAX_SLOTS:
            raise ValueError(f"more than {SEG_MAX_SLOTS} instances in one scan")
        n_labels = max(inst_ids) + 1 if inst_ids else 1
        slot_of_label = np.full((n_labels,), -1, np.int32)
        for s, i in enumerate(inst_ids):
            if i < 0:
                raise ValueError("instance ids must be non-negative")
            slot_of_label[i] = s
        n_slots = max(len(inst_ids), 1)
        sol = torch.as_tensor(slot_of_label).to(self.device)
        offsets = torch.empty((n_slots + 1,), dtype=torch.int32, device=self.device)
        ws = torch.empty((max((n + SEG_CHUNK - 1) // SEG_CHUNK, 1) * n_slots,), dtype=torch.int32,
                         device=self.device)
        self._reserve(n)
        base = self.tail
        with torch.cuda.device(self.device):
            rc = lib.msr3d_segment_scan(n, _p(lab), _p(sol), n_labels, n_slots, _p(pts), _p(col),
                                        _p(self.points[base:]), _p(self.colors[base:]), None, _p(offsets),
                                        _p(ws), _lib.current_stream_ptr(self.device))
        _lib.check(rc, "msr3d_segment_scan")
        off = offsets.cpu().numpy().astype(np.int64)          # load-time sync, once per scan
        keep = [k for k in range(len(inst_ids)) if off[k + 1] > off[k]]
        self.scans[scan_id] = {
            "base": base,
            "inst_ids": [inst_ids[k] for k in keep],
            "begin": {inst_ids[k]: base + int(off[k]) for k in keep},
            "count": {inst_ids[k]: int(off[k + 1] - off[k]) for k in keep},
            "rows": int(off[-1]) if len(inst_ids) else 0,
        }
        self.tail = base + self.scans[scan_id]["rows"]
        return list(self.scans[scan_id]["inst_ids"])

    def add_scan_file(self, scan_id, path, inst_ids=None):
        return self.add_scan(scan_id, *load_scan_pth(path), inst_ids=inst_ids)

    # ------------------------------------------------------------------
    def __contains__(self, scan_id):
        return scan_id in self.scans

    def inst_ids(self, scan_id):
        return self.scans[scan_id]["inst_ids"]

    def obj_pcd(self, scan_id, inst_id):
        """The object's rows as the reference would hold them: float64 (n, 6) [xyz, rgb/127.5-1]
        (debug / test accessor; copies to the host)."""
        s = self.scans[scan_id]
        b, c = s["begin"][inst_id], s["count"][inst_id]
        xyz = self.points[b:b + c].cpu().numpy().astype(np.float64)
        rgb = self.colors[b:b + c].cpu().numpy() / 127.5 - 1
        return np.concatenate([xyz, rgb], 1)

    def nbytes(self):
        return self.tail * 15
