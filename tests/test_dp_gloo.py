"""world_size-2 gloo test (CPU) of the data-parallel exchange step: one sum all-reduce of the flat
live-gradient buffer, 1/world returned as the optimizer's grad_scale (DDP-mean semantics)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from erc_amd.engine import FlatParams, all_reduce_grads
    torch.manual_seed(0)  # identical parameters on every rank
    lin1, lin2 = torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)
    flat = FlatParams([[("a.weight", lin1.weight)], [("a.bias", lin1.bias)],
                       [("b.weight", lin2.weight), ("b.bias", lin2.bias)]], "cpu")
    assert lin1.weight.data_ptr() == flat.w("a.weight").data_ptr()       # parameters are views of the flat buffer
    assert lin2.bias.grad.data_ptr() == flat.g("b.bias").data_ptr()      # and so are their .grad
    # each rank has its own gradient (its own shard of dialogues)
    g = torch.Generator().manual_seed(100 + rank)
    for name in flat.params:
        flat.g(name).copy_(torch.randn(flat.shapes[name], generator=g))
    mine = flat.grad.clone()
    scale = all_reduce_grads(flat)
    others = []
    for r in range(world):
        gg = torch.Generator().manual_seed(100 + r)
        others.append(torch.cat([torch.randn(flat.shapes[n], generator=gg).flatten() for n in flat.params]))
    want_sum = sum(others)
    got = torch.cat([flat.g(n).flatten() for n in flat.params])
    ok = torch.allclose(got, want_sum, atol=1e-6) and abs(scale - 1.0 / world) < 1e-12
    # padding between groups stays zero (the optimizer sweeps the whole buffer)
    live = torch.zeros(flat.numel, dtype=torch.bool)
    for n in flat.params:
        live[flat.offsets[n]:flat.offsets[n] + flat.params[n].numel()] = True
    ok = ok and float(flat.grad[~live].abs().sum()) == 0.0
    # an empty shard still has to enter the collective with zeros
    flat.grad.zero_()
    if rank == 0:
        flat.grad.copy_(mine)
    all_reduce_grads(flat)
    r0 = others[0]
    ok = ok and torch.allclose(torch.cat([flat.g(n).flatten() for n in flat.params]), r0, atol=1e-6)
    # the health word rides behind the gradients: raised on ONE rank (a poll timeout there), non-zero on EVERY rank after
    # the exchange, so all ranks skip the same update (FusedAdam.skip_flag) and all of them report it
    ok = ok and int(flat.health[0]) == 0
    if rank == 1:
        flat.health.fill_(0x3f800000)        # ERC_HEALTH_RAISED
    all_reduce_grads(flat)
    ok = ok and int(flat.health[0]) != 0
    all_reduce_grads(flat)                   # stays non-zero however often it is summed before the next roll
    ok = ok and int(flat.health[0]) != 0 and flat.grad.data_ptr() == flat.grad_full.data_ptr()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_flat_gradient_all_reduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def _loop_worker(rank, world, port, n_dialogs, batch_size, q):
    """The sharding / step-count logic of trainer.run with an odd dialogue count: every rank must issue the same number
    of all-reduces per epoch (one per batch) or the job hangs."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from erc_amd.trainer import ListDataset, shard_dialogues
    from torch.utils.data import DataLoader
    dialogs = [{"id": i} for i in range(n_dialogs)]
    mine = shard_dialogues(dialogs, rank, world)
    loader = DataLoader(ListDataset(mine), batch_size=batch_size, shuffle=True, collate_fn=lambda s: [d for d, in s],
                        generator=torch.Generator().manual_seed(rank))
    steps, seen = 0, []
    for batch in loader:                       # one gradient exchange per batch, as trainer.run does
        t = torch.ones(3)
        dist.all_reduce(t)
        assert float(t[0]) == world
        steps += 1
        seen += [d["id"] for d in batch]
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([steps]))
    q.put((rank, steps, [int(c) for c in counts], sorted(seen)))
    dist.destroy_process_group()


def test_sharded_epoch_runs_equal_steps_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n, bs, world = 259, 43, 2                   # dialogs[rank::world] gave 130 / 129 dialogues -> 4 vs 3 steps
    procs = [ctx.Process(target=_loop_worker, args=(r, world, _free_port_shared(), n, bs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res[0][1] == res[1][1] == 4 and res[0][2] == res[1][2] == [4, 4]
    covered = set(res[0][3]) | set(res[1][3])
    assert covered == set(range(n))             # nothing dropped; exactly one dialogue is seen twice (the wrap)
    assert len(res[0][3]) == len(res[1][3]) == 130


_PORT = []


def _free_port_shared():
    if not _PORT:
        _PORT.append(_free_port())
    return _PORT[0]


def test_shard_dialogues_properties():
    from erc_amd.trainer import shard_dialogues
    for n in (0, 1, 5, 120, 259):
        for world in (1, 2, 3, 8):
            shards = [shard_dialogues(list(range(n)), r, world) for r in range(world)]
            assert len({len(s) for s in shards}) == 1
            if n:
                assert set().union(*map(set, shards)) == set(range(n))
                assert sum(map(len, shards)) - n < world


class _FakeGraph:
    """stands in for a captured HIP graph: ``capture`` records the step WITHOUT executing it, ``replay`` executes it"""

    def __init__(self, fn):
        self.fn = fn

    def replay(self):
        self.fn()


def _graphs_worker(rank, world, port, q):
    """Control flow of trainer.StepGraphs under data parallelism (the captured step contains the gradient all-reduce): every
    rank precaptures all capacity buckets in one order -- each warm-up is a real collective on every rank -- and then walks
    ITS OWN sequence of batches (different buckets per rank and step, some outside every bucket): the number of
    collectives per step is 1 on every rank whatever mix of replay / eager the ranks run, and nothing is captured lazily."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from erc_amd.trainer import StepGraphs
    calls = []

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.bn = torch.nn.BatchNorm1d(3)
            self.dynamic_n = False
            self.flat = types.SimpleNamespace(health=torch.zeros(1, dtype=torch.int32), events=torch.zeros(2, dtype=torch.int32))

    class Trainer:
        model = Model()

        def train_step(self, batch):
            t = torch.ones(2)
            dist.all_reduce(t)                       # the step's one collective
            assert float(t[0]) == world
            calls.append(int(batch["cap"]))
            return torch.zeros(4)

        def _mk(self, cap):
            return (("capacity", cap), (lambda: {"cap": cap, "n": torch.zeros(1)}), (lambda static, b: static["n"].copy_(b["n"])),
                    (lambda static: None))

        def capacity_bucket(self, batch):
            cap = -(-int(batch["n"]) // 4) * 4
            return self._mk(cap)[:3] if cap <= 12 else None

        def all_capacity_buckets(self, probe):
            return [self._mk(c) for c in (4, 8, 12)]

    class Graphs(StepGraphs):
        def _capture(self, fn):
            return _FakeGraph(fn), torch.zeros(4)

        def _sync(self):
            pass

    tr = Trainer()
    g = Graphs(tr)
    g.precapture({"n": torch.tensor([1.0])})
    g.lazy = False
    warm = len(calls)
    sizes = [[3, 9, 30, 5, 12, 1], [11, 2, 7, 40, 4, 8]][rank]            # rank-specific bucket sequences; 30 / 40: no bucket
    for n in sizes:
        g.step({"n": torch.tensor([float(n)]), "cap": -1})
    q.put((rank, warm, len(calls) - warm, g.captures, g.replays, g.eager))
    dist.destroy_process_group()


def test_stepgraphs_data_parallel_control_flow():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graphs_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    # 3 warm-up collectives during precapture, 6 step collectives, 3 graphs, 5 replays + 1 eager (the batch outside every bucket)
    assert res == [(0, 3, 6, 3, 5, 1), (1, 3, 6, 3, 5, 1)]


def test_bench_launches_its_own_ranks():
    """`bench.py --gpus N` without a launcher starts N ranks itself (torch.distributed.run children; the parent touches no
    GPU): under --backend gloo --dry_run the line reports 2 ranks counted by an all-reduce.  With fewer visible devices than
    N (none in this container) the real run refuses with a non-zero return code instead of printing a 1-rank line."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    res = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry_run"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dry_run"]
    import torch
    if torch.cuda.device_count() < 2:
        res = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                             timeout=600, env=env)
        assert res.returncode == 2 and "only" in res.stderr and not res.stdout.strip()
