"""GPU parity of the device-side sample construction (msr3d_segment_scan, msr3d_preprocess_pcd,
msr3d_amd.data) against the vectors from the reference's own functions (tests/golden/
preprocess_seed*.npz) and against the oracle.

Bars: segmentation, masks, colours and drawn indices are integer / byte work -> bit-exact.
xyz and obj_locs are float64 arithmetic cast to fp32 at the end on both sides; the kernel
tree-reduces sums numpy adds sequentially, so values may differ by float64 rounding before the
cast: tolerance 1 fp32 ulp (|a-b| <= 2^-23 * max(|b|, 2^-20)), and at most 1 element in 10^4 may
differ at all."""
import os

import numpy as np
import pytest
import torch

from oracle import sample_input as si

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(seed):
    return dict(np.load(os.path.join(GOLDEN, f"preprocess_seed{seed}.npz"), allow_pickle=False))


def assert_ulp(got, want, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    tol = 2.0 ** -23 * np.maximum(np.abs(want), 2.0 ** -20)
    bad = np.abs(got - want) > tol
    assert not bad.any(), (what, int(bad.sum()), float(np.abs(got - want).max()))
    # (a coincident-point object is ~1e-17 summation noise in numpy and exactly 0 here: not counted)
    differ = ((got != want) & (np.abs(want) > 2.0 ** -30)).mean()
    assert differ <= 1e-4, (what, differ)


def store_with(g, scan_id="scan", inst_ids=None):
    from msr3d_amd.data import SceneStore
    st = SceneStore("cuda", capacity_points=1024)          # small: exercises arena growth
    kept = st.add_scan(scan_id, g["points"], g["colors"], g["instance_labels"], inst_ids=inst_ids)
    return st, kept


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_segmentation_matches_instance_masks(seed):
    g = load(seed)
    st, kept = store_with(g)
    assert kept == g["inst_ids"].tolist()
    order, offsets = si.segment_instances(g["instance_labels"], kept)
    assert st.tail == offsets[-1]
    assert np.array_equal(st.points[:st.tail].cpu().numpy(), g["points"][order])
    assert np.array_equal(st.colors[:st.tail].cpu().numpy(), g["colors"][order])
    pcds = si.scan_to_pcds(g["points"], g["colors"])
    for i in kept[:5]:
        assert np.array_equal(st.obj_pcd("scan", i), pcds[g["instance_labels"] == i])


def test_segmentation_sparse_ids_empty_instances_and_order_output():
    import ctypes
    from msr3d_amd import _lib
    rng = np.random.default_rng(5)
    n = 3001                                                  # not a multiple of the chunk
    ids = [7, 2, 40, 11, 300]                                 # 3RScan-style sparse ids, arbitrary order; 300 is empty
    labels = rng.choice([-100, 0, 2, 7, 11, 40, 41], size=n).astype(np.int64)
    pts = rng.standard_normal((n, 3)).astype(np.float32)
    col = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    from msr3d_amd.data import SceneStore
    st = SceneStore("cuda", capacity_points=16)
    kept = st.add_scan("a", pts, col, labels, inst_ids=ids)
    assert kept == [7, 2, 40, 11]
    order, offsets = si.segment_instances(labels, ids)
    assert np.array_equal(st.points[:st.tail].cpu().numpy(), pts[order])
    # a second scan lands behind the first
    st.add_scan("b", pts[:500], col[:500], labels[:500], inst_ids=[2])
    o2, _ = si.segment_instances(labels[:500], [2])
    b = st.scans["b"]
    assert b["base"] == len(order) and np.array_equal(st.points[b["base"]:st.tail].cpu().numpy(), pts[o2])
    # raw C-ABI call with the optional `order` output; and an all-unlabelled scan
    lib = _lib.load()
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(a).to(dev)                  # noqa: E731
    sol = np.full((301,), -1, np.int32)
    for s, i in enumerate(ids):
        sol[i] = s
    d_order = torch.full((n,), -1, dtype=torch.int32, device=dev)
    d_off = torch.empty((6,), dtype=torch.int32, device=dev)
    ws = torch.empty(((n + 255) // 256 * 5,), dtype=torch.int32, device=dev)
    ps, cs = torch.empty((n, 3), device=dev), torch.empty((n, 3), dtype=torch.uint8, device=dev)
    p = lambda x: ctypes.c_void_p(x.data_ptr())               # noqa: E731
    d = [t(labels), t(sol), t(pts), t(col)]
    rc = lib.msr3d_segment_scan(n, p(d[0]), p(d[1]), 301, 5, p(d[2]), p(d[3]), p(ps), p(cs), p(d_order),
                                p(d_off), p(ws), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_off.cpu().numpy(), offsets)
    assert np.array_equal(d_order[:len(order)].cpu().numpy(), order)
    lab2 = t(np.full((n,), -100, np.int64))
    assert lib.msr3d_segment_scan(n, p(lab2), p(d[1]), 301, 5, p(d[2]), p(d[3]), p(ps), p(cs), None, p(d_off),
                                  p(ws), None) == 0
    assert d_off.cpu().tolist() == [0] * 6
    assert lib.msr3d_segment_scan(n, p(lab2), p(d[1]), 301, 9000, p(d[2]), p(d[3]), p(ps), p(cs), None,
                                  p(d_off), p(ws), None) == -22


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_builder_reproduces_the_reference_sample(seed):
    from msr3d_amd.data import SceneInputBuilder
    g = load(seed)
    st, kept = store_with(g)
    O, P = int(g["max_obj_len"]), int(g["num_points"])
    perm = g["shuffle_perm"].tolist()

    def replay(x):
        x[:] = [x[t] for t in perm]
    sel = si.select_objects(kept, g["scan_insts"].tolist(), O, replay)
    idx = np.zeros((1, O, P), np.int32)
    idx[0, :len(sel)] = g["pcd_idxs"]
    rot = None if g["rot_is_none"] else g["rot_matrix"]
    bld = SceneInputBuilder(st, max_obj_len=O, num_points=P)
    out = bld.build([{"scan_id": "scan", "insts": g["scan_insts"].tolist(),
                      "situation": (g["situation_pos"], g["situation_ori"])}],
                    pcd_idxs=idx, rot_matrices=[rot], selections=[sel])
    fts = out["obj_fts"][0].cpu().numpy()
    assert np.array_equal(out["obj_masks"][0].cpu().numpy(), g["obj_masks"])
    assert out["obj_masks"].dtype == torch.bool
    assert np.array_equal(fts[..., 3:], g["obj_fts"][..., 3:])               # colours + padding: exact
    assert np.array_equal(fts[len(sel):], g["obj_fts"][len(sel):])
    assert_ulp(fts[..., :3], g["obj_fts"][..., :3], "xyz")
    assert_ulp(out["obj_locs"][0].cpu().numpy(), g["obj_locs"], "obj_locs")
    assert np.allclose(out["anchor_locs"][0].cpu().numpy(), g["situation_pos_out"].astype(np.float32), atol=1e-6)
    assert np.allclose(out["anchor_orientation"][0].cpu().numpy(), g["situation_ori_out"].astype(np.float32),
                       atol=1e-6)


def test_selection_uses_python_random_like_the_reference():
    import random
    from msr3d_amd.data import SceneInputBuilder
    g = load(1)
    st, kept = store_with(g)
    bld = SceneInputBuilder(st, max_obj_len=int(g["max_obj_len"]))
    random.seed(123)
    got = bld.select_objects("scan", g["scan_insts"].tolist())
    random.seed(123)
    want = si.select_objects(kept, g["scan_insts"].tolist(), int(g["max_obj_len"]), random.shuffle)
    assert got == want and len(got) == 60 and got[:5] == g["scan_insts"].tolist()


@pytest.mark.parametrize("P", [256, 1024, 2048])
def test_device_drawn_subsample_bit_exact_and_consistent(P):
    """Indices drawn on the device == the oracle's restatement (integer work, bit-exact); the
    sample built from them == the oracle's preprocess_pcd on the same indices."""
    from msr3d_amd.data import SceneInputBuilder
    g = load(0)
    st, kept = store_with(g)
    bld = SceneInputBuilder(st, max_obj_len=32, num_points=P, split="val", seed=0xDEADBEEFCAFE)
    samples = [{"scan_id": "scan", "insts": []}, {"scan_id": "scan", "insts": []}]
    out = bld.build(samples, return_indices=True)
    idx = out["pcd_idxs"].cpu().numpy()
    seed = out["last_seed"]
    pcds = si.scan_to_pcds(g["points"], g["colors"])
    for b in range(2):
        objs = []
        for o, i in enumerate(kept):
            n = st.scans["scan"]["count"][i]
            want = si.draw_indices(seed, b, o, n, P)
            assert np.array_equal(idx[b, o], want), (b, o, n)
            if n >= P:
                assert len(np.unique(idx[b, o])) == P
            objs.append(pcds[g["instance_labels"] == i])
        assert (idx[b, len(kept):] == -1).all()
        fts, locs = si.preprocess_pcd(objs, list(idx[b, :len(kept)]))
        pf, pl, pm = si.pad_sample(fts, locs, 32)
        assert np.array_equal(out["obj_masks"][b].cpu().numpy(), pm)
        got = out["obj_fts"][b].cpu().numpy()
        assert np.array_equal(got[..., 3:], pf[..., 3:])
        assert_ulp(got[..., :3], pf[..., :3], "xyz")
        assert_ulp(out["obj_locs"][b].cpu().numpy(), pl, "locs")
    assert not np.array_equal(idx[0], idx[1])                   # keyed per sample
    again = bld.build(samples, return_indices=True)             # and per step
    assert not np.array_equal(again["pcd_idxs"].cpu().numpy(), idx)


def test_full_batch_properties_and_encoder_consumes_it():
    """BASELINE sizes (16 scenes x 60 objects x 1024 points) from large synthetic scans:
    size-independent properties of the normalisation, determinism, and the encoder runs on it."""
    from msr3d_amd.data import SceneInputBuilder, SceneStore
    from msr3d_amd.synth import synth_scan
    rng = np.random.default_rng(11)
    st = SceneStore("cuda")
    for s in range(4):
        pts, col, lab = synth_scan(rng, 70 + 10 * s, 150000)
        st.add_scan(f"scan{s}", pts, col, lab)
    samples = [{"scan_id": f"scan{b % 4}", "insts": [1, 2, 3],
                "situation": (np.zeros(3), np.array([0, 0, 0, 1.0]))} for b in range(16)]
    bld = SceneInputBuilder(st, seed=3)
    import random
    random.seed(0)
    out = bld.build(samples)
    fts, locs, masks = out["obj_fts"], out["obj_locs"], out["obj_masks"]
    assert fts.shape == (16, 60, 1024, 6) and masks.all()
    xyz = fts[..., :3].double()
    assert xyz.mean(2).abs().max() < 1e-6                        # centred on the subsample mean
    norm = xyz.norm(dim=-1).amax(-1)
    degenerate = norm < 1e-3                                      # the coincident-point object (synth_scan)
    assert ((norm - 1).abs() < 1e-6)[~degenerate].all() and degenerate.sum() >= 1
    assert fts[..., 3:].abs().max() <= 1.0
    assert (locs[..., 3:] >= 0).all()
    random.seed(0)
    bld2 = SceneInputBuilder(st, seed=3)
    out2 = bld2.build(samples)
    assert torch.equal(out2["obj_fts"], fts) and torch.equal(out2["obj_locs"], locs)
    # rotation by 90 degrees about z permutes / negates the box extents and leaves norms alone
    rz = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32)
    sel = [bld.select_objects(s["scan_id"], s["insts"]) for s in samples[:2]]
    idx = np.zeros((2, 60, 1024), np.int32)
    a = bld.build(samples[:2], pcd_idxs=idx, rot_matrices=[None, None], selections=sel)
    r = bld.build(samples[:2], pcd_idxs=idx, rot_matrices=[rz, rz], selections=sel)
    assert torch.allclose(r["obj_locs"][..., 3], a["obj_locs"][..., 4], atol=1e-6)
    assert torch.allclose(r["obj_locs"][..., 0], -a["obj_locs"][..., 1], atol=1e-6)
    # the frozen encoder consumes the batch as it would the loader's
    from tests.helpers import build_prompter
    model = build_prompter("transform", 0, device="cuda")
    tokens = model(dict(out))["obj_tokens"]
    assert tokens.shape == (16, 60, 256) and torch.isfinite(tokens).all()


def test_edge_cases():
    import ctypes
    from msr3d_amd import _lib
    from msr3d_amd.data import SceneInputBuilder, SceneStore
    st = SceneStore("cuda", capacity_points=8)
    pts = np.array([[1, 2, 3], [1, 2, 3], [4, 5, 6], [0, 0, 0]], np.float32)
    col = np.array([[0, 127, 255]] * 4, np.uint8)
    st.add_scan("s", pts, col, np.array([0, 0, 1, 2], np.int64))
    bld = SceneInputBuilder(st, max_obj_len=4, num_points=8, split="val")
    out = bld.build([{"scan_id": "s", "insts": []}])
    fts = out["obj_fts"][0].cpu().numpy()
    assert out["obj_masks"][0].cpu().tolist() == [True, True, True, False]
    assert np.array_equal(fts[:3, :, :3], np.zeros((3, 8, 3)))      # single / coincident points: scale 1
    assert np.array_equal(fts[:3, 0, 3:], np.tile(np.float32(col[0] / 127.5 - 1), (3, 1)))
    assert np.array_equal(fts[3], np.ones((8, 6), np.float32))
    assert np.array_equal(out["obj_locs"][0].cpu().numpy()[:, :3], [[1, 2, 3], [4, 5, 6], [0, 0, 0], [0, 0, 0]])
    empty = bld.build([])
    assert empty["obj_fts"].shape == (0, 4, 8, 6)
    lib = _lib.load()
    assert lib.msr3d_preprocess_pcd(1, 1, 7, None, None, None, None, None, None, ctypes.c_ulonglong(0), None,
                                    None, None, None, None) == -22
    with pytest.raises(ValueError):
        st.add_scan("bad", pts, col.astype(np.float32) + 0.5, np.zeros(4, np.int64))
