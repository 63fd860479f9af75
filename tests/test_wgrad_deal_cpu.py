"""msr3d_wgrad_stream's host side (msr3d_amd/scene_blocks.py::WgradTable._plan_stream): the deal of a launch's tiles, as
one sequence of slab pairs, to a persistent grid.  No device needed."""
import torch


def test_wgrad_stream_deal_covers_every_tile_once_and_cuts_it_at_most_once():
    """The host side of msr3d_wgrad_stream (no launch): every tile's slab range is covered exactly once, a cut tile has
    exactly two parts meeting at an even slab, the parts share a hand-over slot, and no workgroup carries much more than
    its share."""
    from msr3d_amd.scene_blocks import WgradTable
    M, D, FF, W, E, KE = 960, 256, 2048, 816, 4096, 768
    sets = {"bench": [(M, E, D)] + [(M, D, FF), (M, FF, D), (M, W, D), (M, D, D)] * 3 + [(M, D, 64), (M, D, 3), (M, D, KE)],
            "no llm gradient": [(0, E, D)] + [(M, D, FF), (M, FF, D), (M, W, D), (M, D, D)] * 3,
            "as_object": [(976, E, D)] + [(976, D, FF), (976, FF, D), (976, W, D), (976, D, D)] * 3 + [(976, D, 6), (16, D, 84), (976, D, KE)],
            "window of 20": [(1200, E, D)] + [(1200, D, FF), (1200, FF, D), (1200, W, D), (1200, D, D)] * 3 + [(1200, D, KE)]}
    for name, shapes in sets.items():
        for cus in (256, 248):
            t = WgradTable(torch.device("cpu"))
            for m, n, k in shapes:
                t.add(1, n, n, 1, k, k, m, 1, k, 1)
            order, slots, load = t._plan_stream(18, cus=cus, upload=False)
            cov = {}
            for l in order:
                for q in l:
                    if q.kind == 0:
                        cov.setdefault((q.prob, q.ntile, q.ktile), []).append((q.s0, q.s1, q.second, q.slot))
            tiles = sum(-(-p.n_out // 128) * -(-p.k_in // 128) for p in t.probs if p.M > 0)
            assert len(cov) == tiles and sum(1 for l in order for q in l if q.kind == 1) == 18, name
            seen_slots = set()
            for key, v in cov.items():
                v.sort()
                pairs = (t.probs[key[0]].M + 63) >> 6
                if len(v) == 1:
                    assert v[0] == (0, 2 * pairs, 0, -1), (name, key, v)
                else:
                    (a0, a1, asec, aslot), (b0, b1, bsec, bslot) = v
                    assert len(v) == 2 and a0 == 0 and a1 == b0 and b1 == 2 * pairs and a1 % 2 == 0 and 0 < a1 < 2 * pairs
                    assert (asec, bsec) == (0, 1) and aslot == bslot >= 0 and aslot not in seen_slots
                    seen_slots.add(aslot)
            assert len(seen_slots) == slots
            work = sum(((p.M + 63) >> 6) + t.PIECE_CHARGE for p in t.probs if p.M > 0
                       for _ in range(-(-p.n_out // 128) * -(-p.k_in // 128)))
            assert max(load) <= 1.25 * work / len(order) + 2 * t.PIECE_CHARGE, (name, cus, max(load), work / len(order))
