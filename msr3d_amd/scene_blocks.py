"""Host side of the scene-local fused blocks (csrc/scene_block.hip, csrc/wgrad_split.hip): the packed
bf16x3 weight operands and their pack jobs, the launch helper, and the weight-gradient problem table.

A spatial encoder layer (/root/reference/modules/layers/transformers.py:200-252,314-329) needs each of
its weight matrices as the operand of two products -- W x in the forward and dy W in the backward -- so
every matrix is packed twice (plain and transposed), the packed [q|k|v|cond] projection per head
(a head's 32 + 32 + 32 + 6 rows gathered, zero-padded to 128).  msr3d_split_pack does all of it in one
launch from the flat parameter buffer; the packs are valid until the next optimiser step.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import BLK, PRO, PackJob, SceneBlock, SceneRows, WgradProblem

_vp = ctypes.c_void_p
PIECE = 3 * 1024          # bytes of one (slab, tile): three planes of 64 lanes x 16 B


def _device_bytes(ctypes_array, device, pinned=None):
    raw = bytes(ctypes_array)
    host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    if pinned is not None:
        pinned.copy_(host)
        return pinned
    return host.to(device)


class WeightPacks:
    """The packed operands of one model.  `jobs` are fixed (the sources are views of the flat parameter
    buffer); launch() re-splits the current weights."""

    def __init__(self, device):
        self.device = device
        self.jobs = []
        self.bufs = {}
        self.total_pieces = 0
        self._prefix = [0]
        self._table = None

    def add(self, name, src, rows, k, transposed, segs=None, out=None, out_offset=0):
        """Operand `name` (rows x k) from the matrix `src` (2-D f32 view, row stride = src.stride(0))."""
        assert rows % 16 == 0 and k % 32 == 0 and src.dtype == torch.float32 and src.stride(1) == 1
        nbytes = (k // 32) * (rows // 16) * PIECE
        if out is None:
            out = torch.empty(nbytes // 2, dtype=torch.int16, device=self.device)
            self.bufs[name] = out
        j = PackJob()
        j.src, j.ld, j.transposed, j.rows, j.k = src.data_ptr(), src.stride(0), int(transposed), rows, k
        if segs is None:
            segs = [(0, rows if not transposed else k, 0)]
        assert len(segs) <= 4
        j.nseg = len(segs)
        for i, (d, n, s) in enumerate(segs):
            j.seg_dst[i], j.seg_len[i], j.seg_src[i] = d, n, s
        j.dst = out.data_ptr() + out_offset
        self.jobs.append(j)
        self.total_pieces += (k // 32) * (rows // 16)
        self._prefix.append(self.total_pieces)
        self._table = None
        return out

    def nbytes(self, name):
        return self.bufs[name].numel() * 2

    def launch(self, stream, begin=None):
        """begin = (zero_region tensor, n_floats, seed tensor or None): the step's zero fill + seed bump ride as extra
        workgroups of the same launch (msr3d_split_pack_begin)."""
        if self._table is None:
            arr = (PackJob * len(self.jobs))(*self.jobs)
            self._table = _device_bytes(arr, self.device)
            self._pfx = torch.tensor(self._prefix, dtype=torch.int32).to(self.device)
        if begin is not None:
            z, n, seed = begin
            rc = _lib.load().msr3d_split_pack_begin(len(self.jobs), _vp(self._table.data_ptr()), _vp(self._pfx.data_ptr()),
                                                    self.total_pieces, _vp(z.data_ptr()), n,
                                                    _vp(seed.data_ptr()) if seed is not None else None, stream)
            _lib.check(rc, "msr3d_split_pack_begin")
            return
        rc = _lib.load().msr3d_split_pack(len(self.jobs), _vp(self._table.data_ptr()), _vp(self._pfx.data_ptr()),
                                          self.total_pieces, stream)
        _lib.check(rc, "msr3d_split_pack")


def head_segments(h, dh=32, d=256, sd1=6):
    """Rows of the packed [q | k | v | cond] projection (3 d + 8 sd1) that belong to head h, as
    (position in the head's 128-row operand, length, source row)."""
    return [(0, dh, h * dh), (dh, dh, d + h * dh), (2 * dh, dh, 2 * d + h * dh), (3 * dh, sd1, 3 * d + h * sd1)]


HALF_SLOT_FLOATS = 128 * 128 + 128      # MSR3D_WGRAD_HALF_SLOT_FLOATS


class WgradTable:
    """The problems of one msr3d_wgrad_split launch, in device memory; pointers that may move between
    calls (the upstream gradient, the object features) are patched through set_ptr()."""
    TN, TK = 128, 128

    def __init__(self, device):
        self.device = device
        self.probs = []
        self.prefix = [0]
        self._table = None
        self._dirty = True
        self._pin = None
        self._ev = None
        # MSR3D_WGRAD_HALVES=1: every tile's reduction as two units with a ticket hand-over (msr3d_wgrad_split_halves).
        # Built to remove the second, 41 %-full round of the 360 one-per-CU tiles; measured SLOWER in the step (1.306 vs
        # 1.290 ms: two pipeline fills, the parked 64 KB image and the release / acquire fences per tile cost more than the
        # tail they remove), so the default stays one workgroup per tile.  Kept, tested, opt-in.
        self.halves = os.environ.get("MSR3D_WGRAD_HALVES", "0") == "1"
        self._ws = self._sync = None

    def add(self, dy, ldy, n_out, x, ldx, k_in, M, dW, ldw, db):
        p = WgradProblem()
        p.dy, p.ldy, p.n_out, p.x, p.ldx, p.k_in, p.M = dy, ldy, n_out, x, ldx, k_in, M
        p.dW, p.ldw, p.db = dW, ldw, db if db else None
        self.probs.append(p)
        tiles = -(-n_out // self.TN) * -(-k_in // self.TK)
        tiles = (tiles + 7) // 8 * 8          # a multiple of 8 workgroups per problem (XCD-aware tile order)
        self.prefix.append(self.prefix[-1] + tiles)
        self._dirty = True
        return len(self.probs) - 1

    def set_ptr(self, idx, field, ptr):
        if getattr(self.probs[idx], field) != ptr:
            setattr(self.probs[idx], field, ptr)
            self._dirty = True

    def launch(self, stream, colsum=None):
        """colsum = (n_jobs, job-table tensor): msr3d_colsum_partials' jobs as extra workgroups of this launch."""
        if self._dirty:
            arr = (WgradProblem * len(self.probs))(*self.probs)
            n = ctypes.sizeof(arr)
            if self._table is None:
                self._pin = torch.empty(n, dtype=torch.uint8).pin_memory()
                self._table = torch.empty(n, dtype=torch.uint8, device=self.device)
                self._pfx = torch.tensor(self.prefix, dtype=torch.int32).to(self.device)
            if torch.cuda.is_current_stream_capturing():
                # Inside a graph capture the upload becomes a memcpy NODE that re-reads its pinned source at every
                # replay: it gets a staging buffer of its own that is never rewritten (kept alive with the table), and
                # no event of an earlier eager upload is waited for (illegal while capturing).
                pin = torch.empty(n, dtype=torch.uint8).pin_memory()
                self._capture_pins = getattr(self, "_capture_pins", []) + [pin]
                _device_bytes(arr, self.device, pinned=pin)
                self._table.copy_(pin, non_blocking=True)
            else:
                if self._ev is not None:
                    self._ev.synchronize()    # the previous upload has left the pinned staging buffer
                _device_bytes(arr, self.device, pinned=self._pin)
                self._table.copy_(self._pin, non_blocking=True)
                self._ev = torch.cuda.Event()
                self._ev.record()
            self._dirty = False
        if self.halves:
            # every tile's token reduction as two units (2 x tiles workgroups): ~1.4 tiles per CU spread evenly
            # instead of two rounds with a 41 %-full second one; partial hand-over through `_ws`, tickets in `_sync`
            if self._ws is None:
                self._ws = torch.empty(self.prefix[-1] * HALF_SLOT_FLOATS, dtype=torch.float32, device=self.device)
                self._sync = torch.zeros(2 * self.prefix[-1], dtype=torch.int32, device=self.device)
            rc = _lib.load().msr3d_wgrad_split_halves(len(self.probs), _vp(self._table.data_ptr()), _vp(self._pfx.data_ptr()),
                                                      self.prefix[-1], _vp(self._ws.data_ptr()), self._ws.numel(),
                                                      _vp(self._sync.data_ptr()), stream)
            _lib.check(rc, "msr3d_wgrad_split_halves")
            if colsum is not None:
                _lib.check(_lib.load().msr3d_colsum_partials(colsum[0], _vp(colsum[1].data_ptr()), stream), "msr3d_colsum_partials")
            return
        if colsum is not None:
            rc = _lib.load().msr3d_wgrad_split_colsum(len(self.probs), _vp(self._table.data_ptr()), _vp(self._pfx.data_ptr()),
                                                      self.prefix[-1], colsum[0], _vp(colsum[1].data_ptr()), stream)
            _lib.check(rc, "msr3d_wgrad_split_colsum")
            return
        rc = _lib.load().msr3d_wgrad_split(len(self.probs), _vp(self._table.data_ptr()), _vp(self._pfx.data_ptr()),
                                           self.prefix[-1], stream)
        _lib.check(rc, "msr3d_wgrad_split")


def launch_block(stream, **kw):
    s = SceneBlock()
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        elif isinstance(v, _vp):
            v = v.value
        setattr(s, k, v if v is not None else 0)
    rc = _lib.load().msr3d_scene_block(ctypes.byref(s), stream)
    if rc:
        fields = ", ".join(f"{n}={getattr(s, n)!r}" for n, _ in s._fields_ if getattr(s, n))
        _lib.check(rc, f"msr3d_scene_block({fields})")


def launch_rows(stream, **kw):
    s = SceneRows()
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        elif isinstance(v, _vp):
            v = v.value
        setattr(s, k, v if v is not None else 0)
    rc = _lib.load().msr3d_scene_rows(ctypes.byref(s), stream)
    if rc:
        fields = ", ".join(f"{n}={getattr(s, n)!r}" for n, _ in s._fields_ if getattr(s, n))
        _lib.check(rc, f"msr3d_scene_rows({fields})")


__all__ = ["BLK", "PRO", "WeightPacks", "WgradTable", "launch_block", "launch_rows", "head_segments", "PIECE"]
