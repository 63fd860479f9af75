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
from ._lib import BLK, PRO, PackJob, SceneBlock, SceneRows, WgradPiece, WgradProblem

_vp = ctypes.c_void_p
PIECE = 3 * 1024          # bytes of one (slab, tile): three planes of 64 lanes x 16 B

# Arithmetic of the blocks' and the weight gradients' products:
#   "f32"   (default) every product as six bf16 MFMA products of exactly split operands: fp32 accuracy
#   "bf16"  LABELLED reduced variant (MSR3D_TRAIN_MMA=bf16): operands rounded to bf16, ONE MFMA product per product, fp32
#           accumulate, fp32 storage of every activation -- what `torch.autocast(bfloat16)` linears compute; the kernels are
#           the same sources compiled with MSR3D_TRAIN_PLANES=1 (libmsr3d_hip_bf16.so).  Never the headline.
_TRAIN_MMA = [os.environ.get("MSR3D_TRAIN_MMA", "f32")]
if _TRAIN_MMA[0] not in ("f32", "bf16"):
    raise ValueError("MSR3D_TRAIN_MMA must be 'f32' or 'bf16'")


def train_mma():
    return _TRAIN_MMA[0]


def set_train_mma(name):
    if name not in ("f32", "bf16"):
        raise ValueError("train mma must be 'f32' or 'bf16'")
    prev, _TRAIN_MMA[0] = _TRAIN_MMA[0], name
    return prev


def set_attn_fwd_form(split):
    """The attention forward block as two workgroups per (scene, head) (True: the library's default, best when the step has
    the chip to itself) or one (False: best in the pipelined schedule, where the next batch's encoder runs beside the
    trainable part); applies to every loaded library; takes effect at the next launch (a captured graph keeps its form)."""
    form = 1 if split else 0
    for lib in (_lib.load(), _lib._lib_bf16):
        if lib is not None:
            lib.msr3d_attn_fwd_form(form)


def _klib():
    """The library whose block / weight-gradient kernels run."""
    return _lib.load_bf16() if _TRAIN_MMA[0] == "bf16" else _lib.load()


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
        # Round 5 (default; MSR3D_WGRAD_MIXED=0 turns it off): only the tiles of the launch's PARTIAL round are cut in two
        # (msr3d_wgrad_split_mixed) -- ~330 real tiles on 256 CUs are one full round and one that is 29 % full; with those
        # 74 tiles as 148 half-reductions the second round lasts half a tile time.
        self.mixed = os.environ.get("MSR3D_WGRAD_MIXED", "1") != "0" and not self.halves
        self._ws = self._sync = None
        self._real = []                       # real (non-padding) tiles of each problem
        # Round 6 (default; MSR3D_WGRAD_STREAM=0 turns it off): the tiles as one sequence of slab pairs dealt evenly to a
        # persistent grid, one workgroup per CU, cut tiles completed by a small second launch (msr3d_wgrad_stream).  Alone
        # it is EQUAL to the mixed launch (96.7-100.7 us against 98.5-99.8, same box, tools/bench_wgrad.py: ~2.5 pieces
        # per workgroup each pay a prologue and a write-back, which is what the even deal saves); IN THE STEP it is 8-12 us
        # faster (0.851 against 0.862 ms, three interleaved pairs: profiles/r06_v2_ab.txt).  Bit-reproducible.  Needs the
        # pipe tile kernel (msr3d_wgrad_form 1).
        self.stream = os.environ.get("MSR3D_WGRAD_STREAM", "1") != "0" and not self.halves
        self._stream_key = None
        self._stream_plan = None

    def add(self, dy, ldy, n_out, x, ldx, k_in, M, dW, ldw, db):
        p = WgradProblem()
        p.dy, p.ldy, p.n_out, p.x, p.ldx, p.k_in, p.M = dy, ldy, n_out, x, ldx, k_in, M
        p.dW, p.ldw, p.db = dW, ldw, db if db else None
        self.probs.append(p)
        tiles = -(-n_out // self.TN) * -(-k_in // self.TK)
        self._real.append(tiles)
        tiles = (tiles + 7) // 8 * 8          # a multiple of 8 workgroups per problem (XCD-aware tile order)
        self.prefix.append(self.prefix[-1] + tiles)
        self._dirty = True
        return len(self.probs) - 1

    def _balance_xcds(self):
        """xcd_rot of every problem: workgroup b runs on XCD b % 8 and a problem's real tiles occupy the first
        ceil-runs of its 8 XCD slots, so without a rotation the padding always falls on the high XCDs.  Greedy: problems
        in order, each takes the rotation that leaves the fullest XCD emptiest."""
        load = [0] * 8
        for p, r, a, b in zip(self.probs, self._real, self.prefix[:-1], self.prefix[1:]):
            nb = b - a
            per = [sum(1 for lb in range(x, nb, 8) if x * (nb >> 3) + (lb >> 3) < r) for x in range(8)]   # slot x' -> tiles
            if p.M <= 0:
                per = [0] * 8
            best = min(range(8), key=lambda rot: (max(load[(x + rot) & 7] + per[x] for x in range(8)), rot))
            p.xcd_rot = best
            for x in range(8):
                load[(x + best) & 7] += per[x]
        self.xcd_load = load

    def _whole_tiles(self):
        """Workgroups [0, W) of the mixed launch take whole tiles, the rest run as two half-reductions each (see below).
        Problems whose token count is zero this step still own their workgroups."""
        key = tuple(p.M for p in self.probs)
        if getattr(self, "_whole_key", None) == key:
            return self._whole
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        real = []                                          # per workgroup slot: does it multiply?
        for p, r, a, b in zip(self.probs, self._real, self.prefix[:-1], self.prefix[1:]):
            nb = b - a
            for lb in range(nb):
                local = ((lb - p.xcd_rot) & 7) * (nb >> 3) + (lb >> 3)     # (the kernel's XCD-aware order)
                real.append(local < r and p.M > 0)
        # Measured (tools/bench_wgrad.py, the step's 326 real tiles of 360 workgroups on 256 CUs): whole tiles only 115 us,
        # every tile halved 129 us, the last 96-136 workgroups halved 110 us -- a unit costs ~27 us besides its slabs
        # (its life is set by the loader waves: 16 waves of splitting VALU + MFMA issue on one CU), so cutting helps only
        # the tail.  W: the real tiles left whole number one per CU less a sixteenth (the halves then start on CUs that
        # are already free); nothing is cut when the tiles fit one round or the rest would not fit a second.
        total_real, whole = sum(real), self.prefix[-1]
        target = cus - cus // 16
        if total_real > cus and 2 * (total_real - target) <= cus:
            seen, w = 0, 0
            while w < self.prefix[-1] and seen + sum(real[w:w + 8]) <= target:
                seen += sum(real[w:w + 8])
                w += 8
            whole = w
        self._whole_key, self._whole = key, whole
        return whole

    PIECE_CHARGE = 2          # what a piece costs besides its slab pairs (prologue + write-back), in slab pairs

    def _plan_stream(self, n_jobs, cus=None, upload=True):
        """-> (pieces tensor, wg_first tensor, n_pieces, n_wgs, n_slots) for msr3d_wgrad_stream, or None when the problem set
        is too small to deal (fewer slab pairs than one tile's worth per workgroup: the plain launch is already one round).
        Rebuilt when a problem's token count changes (llm_proj's is 0 in a step without an upstream gradient)."""
        key = (tuple((p.M, p.n_out, p.k_in) for p in self.probs), n_jobs)
        if self._stream_key == key:
            return self._stream_plan
        C = self.PIECE_CHARGE
        tiles = []                           # (problem, ntile, ktile, slab pairs): consecutive k tiles share a dy block
        for pi, p in enumerate(self.probs):
            if p.M <= 0:
                continue
            pairs = (p.M + 63) >> 6
            for nt in range(-(-p.n_out // self.TN)):
                for kt in range(-(-p.k_in // self.TK)):
                    tiles.append((pi, nt, kt, pairs))
        plan = None
        if cus is None:
            cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        if tiles:
            total = sum(t[3] + C for t in tiles)
            heaviest = max(t[3] + C for t in tiles)
            n_wgs = min(cus, total // heaviest)       # every workgroup's share >= the heaviest tile: a tile is cut at most once
            if n_wgs >= 8 and len(tiles) > n_wgs // 2:
                n_wgs -= n_wgs % 8
                lists = [[] for _ in range(n_wgs)]
                slots, c, filled, rem = 0, 0, 0.0, float(total + C * (n_wgs - 1))   # (+ the charge of a cut per chunk boundary)
                target = rem / n_wgs                 # re-derived at every chunk start from what is left (cuts add charges)

                def next_chunk():
                    nonlocal c, filled, target
                    c += 1
                    filled = 0.0
                    target = rem / (n_wgs - c)

                for pi, nt, kt, pairs in tiles:
                    w = pairs + C
                    while True:
                        room = target - filled
                        if c == n_wgs - 1 or w <= room + 0.5:          # the whole tile fits this chunk (or it is the last)
                            lists[c].append(WgradPiece(0, pi, nt, kt, 0, 2 * pairs, 0, -1))
                            filled += w
                            rem -= w
                            break
                        first = int(round(room - C))                    # slab pairs of the part that stays in this chunk
                        if pairs < 2 or first < 1:                      # nothing worth leaving behind: the tile opens the next
                            next_chunk()
                            continue
                        first = min(first, pairs - 1)
                        lists[c].append(WgradPiece(0, pi, nt, kt, 0, 2 * first, 0, slots))
                        rem -= first + C
                        next_chunk()
                        lists[c].append(WgradPiece(0, pi, nt, kt, 2 * first, 2 * pairs, 1, slots))
                        slots += 1
                        filled += pairs - first + C
                        rem -= pairs - first + C
                        break
                    if filled >= target - 0.5 and c < n_wgs - 1:
                        next_chunk()
                load = [sum((q.s1 - q.s0) // 2 + C for q in l) for l in lists]
                for j in range(n_jobs):                                # column-sum jobs: to the lightest workgroups
                    k = min(range(n_wgs), key=lambda i: load[i])
                    lists[k].append(WgradPiece(1, j, 0, 0, 0, 0, 0, -1))
                    load[k] += 1
                # chunk -> workgroup: consecutive chunks on ONE XCD (workgroup b runs on XCD b % 8), so the tiles that share a
                # dy block read it through one L2
                per = n_wgs // 8
                order = [None] * n_wgs
                for ch in range(n_wgs):
                    order[(ch % per) * 8 + ch // per] = lists[ch]
                flat, first_idx = [], [0]
                for l in order:
                    flat.extend(l)
                    first_idx.append(len(flat))
                slot_piece = [0] * slots
                for i, q in enumerate(flat):
                    if q.kind == 0 and q.slot >= 0 and not q.second:
                        slot_piece[q.slot] = i
                arr = (WgradPiece * len(flat))(*flat)
                if not upload:                      # (tests of the deal itself, no device)
                    return order, slots, load
                plan = (_device_bytes(arr, self.device), torch.tensor(first_idx, dtype=torch.int32).to(self.device),
                        len(flat), n_wgs, slots, max(load), min(load),
                        torch.tensor(slot_piece or [0], dtype=torch.int32).to(self.device))
        self._stream_key, self._stream_plan = key, plan
        return plan

    def set_ptr(self, idx, field, ptr):
        if getattr(self.probs[idx], field) != ptr:
            setattr(self.probs[idx], field, ptr)
            self._dirty = True

    def launch(self, stream, colsum=None):
        """colsum = (n_jobs, job-table tensor): msr3d_colsum_partials' jobs as extra workgroups of this launch."""
        if self._dirty:
            self._balance_xcds()
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
            rc = _klib().msr3d_wgrad_split_halves(len(self.probs), _vp(self._table.data_ptr()), _vp(self._pfx.data_ptr()),
                                                      self.prefix[-1], _vp(self._ws.data_ptr()), self._ws.numel(),
                                                      _vp(self._sync.data_ptr()), stream)
            _lib.check(rc, "msr3d_wgrad_split_halves")
            if colsum is not None:
                _lib.check(_lib.load().msr3d_colsum_partials(colsum[0], _vp(colsum[1].data_ptr()), stream), "msr3d_colsum_partials")
            return
        if self.stream and _klib().msr3d_wgrad_form(-1) == 1:
            if torch.cuda.is_current_stream_capturing() and self._stream_key != (
                    tuple((p.M, p.n_out, p.k_in) for p in self.probs), colsum[0] if colsum is not None else 0):
                raise RuntimeError("WgradTable: the stream plan must exist before a graph capture (launch once eagerly)")
            plan = self._plan_stream(colsum[0] if colsum is not None else 0)
            if plan is not None:
                pieces, first, n_pieces, n_wgs, slots = plan[:5]
                if slots and (self._ws is None or self._ws.numel() < slots * HALF_SLOT_FLOATS):
                    if self._ws is not None and torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("WgradTable: the workspace would have to grow inside a graph capture")
                    T = max(self.prefix[-1], slots)
                    self._ws = torch.empty(T * HALF_SLOT_FLOATS, dtype=torch.float32, device=self.device)
                    self._sync = torch.zeros(2 * T, dtype=torch.int32, device=self.device)
                rc = _klib().msr3d_wgrad_stream(len(self.probs), _vp(self._table.data_ptr()), n_pieces, _vp(pieces.data_ptr()),
                                                _vp(first.data_ptr()), n_wgs, slots, _vp(plan[7].data_ptr()),
                                                _vp(self._ws.data_ptr()) if slots else None,
                                                self._ws.numel() if slots else 0,
                                                _vp(colsum[1].data_ptr()) if colsum is not None else None, stream)
                _lib.check(rc, "msr3d_wgrad_stream")
                return
        whole = self._whole_tiles() if self.mixed else self.prefix[-1]
        if whole < self.prefix[-1]:
            H = self.prefix[-1] - whole
            if self._ws is None or self._ws.numel() < H * HALF_SLOT_FLOATS:
                # sized ONCE for the worst case (every tile halved), so that a later launch whose problems' token counts
                # -- and with them the whole / halved cut -- differ never moves the buffers a captured graph points at
                if self._ws is not None and torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("WgradTable: the hand-over workspace would have to grow inside a graph capture "
                                       "(problems were added after the table's first launch)")
                T = self.prefix[-1]
                self._ws = torch.empty(T * HALF_SLOT_FLOATS, dtype=torch.float32, device=self.device)
                self._sync = torch.zeros(2 * T, dtype=torch.int32, device=self.device)
            nj, jt = (colsum[0], _vp(colsum[1].data_ptr())) if colsum is not None else (0, _vp(0))
            rc = _klib().msr3d_wgrad_split_mixed(len(self.probs), _vp(self._table.data_ptr()), _vp(self._pfx.data_ptr()),
                                                     self.prefix[-1], whole, nj, jt, _vp(self._ws.data_ptr()),
                                                     self._ws.numel(), _vp(self._sync.data_ptr()), stream)
            _lib.check(rc, "msr3d_wgrad_split_mixed")
            return
        if colsum is not None:
            rc = _klib().msr3d_wgrad_split_colsum(len(self.probs), _vp(self._table.data_ptr()), _vp(self._pfx.data_ptr()),
                                                      self.prefix[-1], colsum[0], _vp(colsum[1].data_ptr()), stream)
            _lib.check(rc, "msr3d_wgrad_split_colsum")
            return
        rc = _klib().msr3d_wgrad_split(len(self.probs), _vp(self._table.data_ptr()), _vp(self._pfx.data_ptr()),
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
    rc = _klib().msr3d_scene_block(ctypes.byref(s), stream)
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
