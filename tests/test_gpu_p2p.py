"""ERC_DP_P2P=1: the gradient exchange fused into the optimizer launch (csrc/optim.hip P2PArgs, engine.P2PExchange).
Functional check with TWO PROCESSES SHARING THE ONE GPU of the box: each maps the other's publish buffer and flag array
through hipIpc handles (exchanged over a gloo group) and runs erc_adam_step_p2p on its own, different gradient.  Both must
end with the parameters torch.optim.Adam gives for the MEAN gradient, bit-identical between the ranks; a health bit raised
on one rank makes both skip that step.  (xGMI between distinct devices is not exercised: one-GPU box.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _grads(step, rank, n):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return torch.randn(n, generator=g)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ERC_DP_P2P="1", ERC_P2P_SPIN="400000",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from erc_amd import capi
        from erc_amd.engine import FlatParams, FusedAdam
        torch.cuda.set_device(0)
        torch.manual_seed(0)
        lin1, lin2 = torch.nn.Linear(300, 200), torch.nn.Linear(200, 7)          # 61 607 live parameters: 61 workgroups
        flat = FlatParams([[("a.weight", lin1.weight)], [("a.bias", lin1.bias)], [("b.weight", lin2.weight), ("b.bias", lin2.bias)]],
                          "cuda:0")
        opt = FusedAdam(flat, lr=1e-2, weight_decay=1e-3)
        assert opt.enable_p2p()
        w0 = flat.data.clone()
        skipped_at = 2
        for step in range(4):
            flat.grad.copy_(_grads(step, rank, flat.numel).cuda())
            if step == skipped_at and rank == 1:
                flat.health.fill_(capi.HEALTH_RAISED)        # this rank's gradients are invalid: BOTH ranks must skip
            opt.step(grad_scale=1.0 / world)
            torch.cuda.synchronize()
            if step == skipped_at:
                flat.health.zero_()
        q.put((rank, flat.data.cpu().numpy(), int(opt.state[0]), w0.cpu().numpy(), int(flat.health[0])))   # (numpy: pickled by value)
        dist.barrier()
        flat.p2p.close()
    finally:
        dist.destroy_process_group()


def test_fused_p2p_exchange_two_processes_one_gpu():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    (_, w_a, steps_a, w0, h_a), (_, w_b, steps_b, _, h_b) = res
    w_a, w_b, w0 = torch.from_numpy(w_a), torch.from_numpy(w_b), torch.from_numpy(w0)
    assert steps_a == steps_b == 3 and h_a == 0 and h_b == 0          # 4 launches, one skipped on both ranks
    assert torch.equal(w_a, w_b)                                       # rank-ordered sums: bit-identical replicas
    # reference: torch.optim.Adam on the mean gradient of the three applied steps
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.Adam([p], lr=1e-2, weight_decay=1e-3)
    for step in (0, 1, 3):
        p.grad = (_grads(step, 0, w0.numel()) + _grads(step, 1, w0.numel())) / 2
        opt.step()
    assert float((w_a - p.detach()).abs().max()) < 2e-6


def _trainer_worker(rank, world, port, q, compute):
    """COGMENTrainer under ERC_DP_P2P=1: the exchange rides INSIDE the weight-gradient + optimizer launch (erc_wgrad_adam_p2p)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ERC_DP_P2P="1", ERC_P2P_SPIN="4000000",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import track_mm.cogmen as plugin
        from tests.util_cases import cogmen_case
        torch.cuda.set_device(0)
        params = plugin.ParamsType().from_args(["--dataset=iemocap-cogmen-6", "--modality=atv", "--compute=" + compute, "--seed=3"])
        params.optim.lr = 1e-3
        tr = plugin.COGMENTrainer(params, "cuda:0")
        tr.model.drop_p = 0.0
        assert tr.model.flat.p2p is not None
        w0 = tr.model.flat.data.clone()
        grads = []
        for step in range(3):       # every rank its own batch (data parallel), same shapes on both (same launch table)
            b = tr.prepare_batch(cogmen_case(B=6, min_len=30, max_len=30, dims=params.dims(), seed=100 + 10 * step + rank,
                                             n_classes=params.n_classes)["batch"])
            tr.train_step(b)
            torch.cuda.synchronize()
            assert tr.model._last_ws["planner"].adam_fused
            grads.append(tr.model.flat.grad.cpu().numpy().copy())       # the launch stores the rank-ordered SUM
        q.put((rank, tr.model.flat.data.cpu().numpy(), int(tr.optim.state[0]), w0.cpu().numpy(), int(tr.model.flat.health[0]), grads))
        dist.barrier()
        tr.model.flat.p2p.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("compute", ["bf16", "f32x32"])
def test_exchange_inside_the_weight_gradient_launch_two_processes_one_gpu(compute):
    """Two COGMEN trainers (two processes on the one GPU, ERC_DP_P2P=1) on DIFFERENT batches of the same shape: the step is the
    single-rank 5-launch step -- the weight-gradient launch's work items exchange their summed quads with the same item of the
    peer and apply Adam with the rank-ordered sum x 1/2.  Both replicas end bit-identical, the stored gradient is the same sum on
    both, three steps were counted, no health event; and the parameters are what torch.optim.Adam gives for the MEAN of the two
    ranks' gradients (the ranks' local gradients recomputed by single-process trainers on the same batches)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q, compute)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    (_, w_a, steps_a, w0, h_a, g_a), (_, w_b, steps_b, _, h_b, g_b) = res
    assert steps_a == steps_b == 3 and h_a == 0 and h_b == 0
    assert (w_a == w_b).all()
    for x, y in zip(g_a, g_b):
        assert (x == y).all()
    # reference: the same three steps as plain Adam on half the exchanged sum
    w0 = torch.from_numpy(w0)
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.Adam([p], lr=1e-3, weight_decay=1e-8)
    for g in g_a:
        p.grad = torch.from_numpy(g) * 0.5
        opt.step()
    live = torch.from_numpy(w_a) != w0
    assert float((torch.from_numpy(w_a) - p.detach()).abs().max()) < 2e-6 and int(live.sum()) > 1000
