"""N>1 path on CPU: two gloo ranks run the flat-buffer gradient exchange
(msr3d_amd/dp.py) and must end with identical, correctly averaged gradients -- including
for a parameter that receives NO gradient on one or both ranks (what the reference needs
find_unused_parameters=True for, leo_trainer.py:50)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.unused = torch.nn.Parameter(torch.ones(5))
        self.b = torch.nn.Linear(16, 4)
        self.frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from msr3d_amd.dp import FlatGradAllReduce
    torch.manual_seed(0)
    model = Toy()
    eng = FlatGradAllReduce(model.parameters(), bucket_bytes=256, overlap=(rank >= 0 and os.environ.get("MSR3D_TEST_OVERLAP", "1") == "1"))   # several small buckets
    assert len(eng.buckets) > 1
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    torch.manual_seed(100 + rank)
    for step in range(3):
        x = torch.randn(6, 8)
        eng.zero_grad()
        loss = model(x).pow(2).mean()
        loss.backward()
        local = {n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad}
        # recompute what the local gradient was before the exchange for the check below
        eng.finish()
        eng.clip_grad_norm_(5.0)
        opt.step()
    out = {n: p.detach().numpy().copy() for n, p in model.named_parameters()}   # numpy: pickled by value
    out["__flat__"] = eng.flat.numpy().copy()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_gloo_gradient_exchange(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # identical weights and identical (averaged) flat gradients on both ranks after 3 steps
    import numpy as np
    for r in range(1, world):
        for k in res[0]:
            assert np.allclose(res[0][k], res[r][k], atol=1e-7), (r, k)
    assert np.count_nonzero(res[0]["__flat__"]) > 0
    # the unused parameter's gradient stayed exactly zero and the parameter only saw weight decay
    assert res[0]["unused"].max() < 1.0 and np.allclose(res[0]["unused"], res[0]["unused"][0])


def test_single_process_average_matches_manual():
    """world=1: flat views receive exactly what autograd produces; clip scales globally."""
    from msr3d_amd.dp import FlatGradAllReduce
    torch.manual_seed(0)
    m = Toy()
    eng = FlatGradAllReduce(m.parameters(), bucket_bytes=1 << 20)
    x = torch.randn(4, 8)
    eng.zero_grad()
    (m(x).sum() * 100).backward()
    eng.finish()
    ref = torch.autograd.grad((m(x).sum() * 100), [m.a.weight, m.b.weight])
    assert torch.allclose(m.a.weight.grad, ref[0]) and torch.allclose(m.b.weight.grad, ref[1])
    n = eng.clip_grad_norm_(5.0)
    assert n > 5.0 and abs(eng.grad_norm().item() - 5.0) < 1e-3
    assert m.a.weight.grad.data_ptr() >= eng.flat.data_ptr()      # still a view of the flat buffer


class Twice(torch.nn.Module):
    """`enc` is applied twice in one forward (like loc_embedding_encoder in 'as_embedding'): its
    gradient hook fires once, a direct-write producer would report it twice."""

    def __init__(self):
        super().__init__()
        self.enc = torch.nn.Linear(8, 8)
        self.head = torch.nn.Linear(8, 4)

    def forward(self, x):
        return self.head(torch.tanh(self.enc(x)) + self.enc(2 * x))


def _data(rank, micro):
    g = torch.Generator().manual_seed(1000 + 10 * rank + micro)
    return torch.randn(5, 8, generator=g)


def _worker_accum(rank, world, port, q, accum):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from msr3d_amd.dp import FlatGradAllReduce
    torch.manual_seed(rank)                      # rank-dependent init: the engine must broadcast rank 0's
    model = Twice()
    eng = FlatGradAllReduce(model.parameters(), bucket_bytes=64, overlap=True)
    assert len(eng.buckets) > 1
    _, spread = eng.replica_checksum()
    assert spread == 0.0
    eng.zero_grad()
    for micro in range(accum):
        eng.begin_micro(last=micro + 1 == accum)
        (model(_data(rank, micro)).pow(2).mean() / accum).backward()
        for p in model.enc.parameters():         # a producer reporting the same parameter again
            eng.mark_ready(p)
        if micro + 1 < accum:
            assert not any(eng._launched), "a bucket was exchanged before the last micro-batch"
    eng.finish()
    q.put((rank, eng.flat.numpy().copy(), [p.detach().numpy().copy() for p in eng.order]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("accum", [1, 3])
def test_overlap_with_accumulation_exchanges_once_and_params_are_broadcast(accum):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_accum, args=(r, world, port, q, accum)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (f, ps) for r, f, ps in (q.get(timeout=180) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    # reference: rank 0's weights, gradient = mean over ranks of the sum over micro-batches
    from msr3d_amd.dp import FlatGradAllReduce
    torch.manual_seed(0)
    model = Twice()
    eng = FlatGradAllReduce(model.parameters(), bucket_bytes=64)
    eng.zero_grad()
    for r in range(world):
        for micro in range(accum):
            (model(_data(r, micro)).pow(2).mean() / accum / world).backward()
    want = eng.flat.numpy()
    for r in range(world):
        assert np.allclose(res[r][0], want, rtol=1e-5, atol=1e-7), r
        for a, b in zip(res[r][1], [p.detach().numpy() for p in eng.order]):
            assert np.array_equal(a, b)          # every rank holds rank 0's initial weights


def test_probe_unused_finds_parameters_without_a_gradient_and_offsets_are_aligned():
    """`probe_unused`: one forward/backward under hooks (+ the direct producers' mark_ready) names the
    parameters that receive no gradient -- what FlatAdamW then leaves untouched, as torch's AdamW leaves
    `.grad is None`; every parameter starts on a 16-byte boundary of the flat buffer."""
    from msr3d_amd.dp import FlatGradAllReduce
    torch.manual_seed(0)
    model = Toy()
    eng = FlatGradAllReduce(model.parameters())
    x = torch.randn(4, 8)

    def run():
        eng.zero_grad()
        model(x).sum().backward()
    unused = eng.probe_unused(run)
    assert len(unused) == 1 and unused[0] is model.unused
    # a producer that writes the flat buffer itself reports through mark_ready
    unused = eng.probe_unused(lambda: eng.mark_ready(model.unused))
    assert {id(p) for p in unused} == {id(p) for p in (model.a.weight, model.a.bias, model.b.weight, model.b.bias)}
    assert all(eng.offset[id(p)] % 4 == 0 for p in eng.order)
    assert eng.flat.numel() % 32 == 0
    # the hooks are gone afterwards
    run()
    assert eng._probe is None


def _worker_sum(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from msr3d_amd.dp import FlatGradAllReduce
    torch.manual_seed(0)
    outs = []
    for in_opt in (False, True):
        torch.manual_seed(0)
        model = Twice()
        eng = FlatGradAllReduce(model.parameters(), bucket_bytes=64)
        eng.scale_in_optimizer = in_opt        # the consumer (FlatAdamW) reads g * (1 / world): the engine leaves the SUM
        eng.zero_grad()
        model(_data(rank, 0)).pow(2).mean().backward()
        eng.finish()
        outs.append(eng.flat.numpy().copy())
    q.put((rank, outs))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_leaves_the_sum_when_the_optimiser_does_the_averaging():
    """dp.scale_in_optimizer (HotPathTrainStep sets it with the fused optimiser): after the exchange the buffer holds
    the all-reduced SUM -- world x the mean the default leaves -- on every rank."""
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sum, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        mean, total = res[r]
        assert np.abs(mean).max() > 0
        assert np.allclose(total, world * mean, rtol=1e-6, atol=1e-7)
    assert np.array_equal(res[0][1], res[1][1])


# ---------------------------------------------------------------------------------------------------------
# HotPathTrainStep._capture_checked: every rank issues the same collectives whatever happens to it locally.
# The capture itself needs a GPU; its control flow does not: a step object whose capture / replay / eager
# phases are stand-ins, rank 1's capture raising -> BOTH ranks fall back (no rank left inside a collective).
# ---------------------------------------------------------------------------------------------------------
def _checked_worker(rank, world, port, q, fail_rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from msr3d_amd.train_step import HotPathTrainStep

    class _DP:
        distributed, group, defer_comm = True, None, False

        def replica_checksum(self, t):
            s = t.double().sum().reshape(1)
            lo, hi = s.clone(), s.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            return float(s), float(hi - lo)

    calls = []
    step = HotPathTrainStep.__new__(HotPathTrainStep)
    step.dp, step.static, step.graph, step.graph_comm_check = _DP(), {"obj_embeds": torch.zeros(1)}, None, None
    state = torch.zeros(4)

    class _Graph:
        def replay(self):
            calls.append("replay")
            t = torch.ones(1)
            dist.all_reduce(t)                       # (a captured collective runs on replay)
            state.add_(1.0)

    def capture_graph(batch):
        calls.append("capture" if step._graph_comm else "capture_eager")
        if step._graph_comm and rank == fail_rank:
            raise RuntimeError("capture failed on this rank only")
        step.graph = _Graph()

    def train_part():
        calls.append("eager")
        t = torch.ones(1)
        dist.all_reduce(t)
        state.add_(1.0)

    step._snapshot = lambda: state.clone()
    step._restore = lambda snap: state.copy_(snap)
    step._state_vector = lambda: state.clone()
    step._capture_graph = capture_graph
    step._load = lambda batch: None
    step._train_part = train_part
    step._capture_checked(None)
    q.put((rank, dict(step.graph_comm_check), calls, step._graph_comm))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_capture_check_phases_agree_across_ranks(fail_rank):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_checked_worker, args=(r, world, port, q, fail_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, chk, calls, gc = q.get(timeout=120)      # a hang (mismatched collectives) fails here
        res[rank] = (chk, calls, gc)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        chk, calls, gc = res[rank]
        if fail_rank < 0:
            assert chk["captured"] and gc and calls.count("replay") == 2 and calls.count("eager") == 2
            assert chk["replica_checksum_spread"] == 0.0 and chk["max_abs_diff"] == 0.0
        else:
            # nobody replayed (the failing rank never held a graph), everybody re-captured for the eager exchange
            assert not chk["captured"] and not gc and "replay" not in calls and calls[-1] == "capture_eager"
    if fail_rank >= 0:
        assert "capture failed" in res[fail_rank][0]["why"] and "another rank" in res[1 - fail_rank][0]["why"]
