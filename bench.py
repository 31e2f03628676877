#!/usr/bin/env python3
"""Headline benchmark: valid utterances / second of one COGMEN training step
(graph construction + forward + cross entropy + backward + Adam), BASELINE.json
config 2: iemocap-cogmen-6 shaped a+t+v features with d_a=100, d_t=768, d_v=512
(D=1380), B=32 dialogues per GPU, T=110, 6 classes, synthetic data, feature
block stored in bf16 (``--dtype f32`` runs the fp32 parity path instead).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU,
RCCL): dialogues are sharded data-parallel, per-GPU batch fixed (weak scaling),
one sum all-reduce of the flat live-gradient buffer per step.

One JSON line on rank 0; see DESIGN.md "Measurement" for the roofline and
cpu_baseline definitions.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def synthetic_batch(params, B, T, seed):
    from erc_amd.collate import ERCCollate
    from erc_amd.synthetic import make_dialogues
    dialogs = make_dialogues(B, params.dims(), n_speakers=params.n_speakers, n_classes=params.n_classes,
                             min_len=20, max_len=T, seed=seed, force_max=True)
    batch = ERCCollate(params)([[d] for d in dialogs])
    batch.pop("utterance_texts", None)
    return batch


def cpu_baseline(params, batch, budget_s=20.0):
    """The reference CPU path = the oracle (structure-faithful PyTorch-CPU restatement, dead encoder and
    per-edge python graph construction included) timed on this box's host cores, on a bounded sample."""
    from oracle.cogmen import COGMENOracle, cogmen_train_step
    n_utt = int(batch["label"].shape[0])
    out = {}
    for tag, dead in (("with_dead_encoder", True), ("without_dead_encoder", False)):
        torch.manual_seed(1)
        model = COGMENOracle(params.hidden_all, 100, 17, params.n_speakers, params.n_classes, dead_encoder=dead)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-8)
        cogmen_train_step(model, opt, batch)  # warm-up
        times = []
        t_all = time.perf_counter()
        while len(times) < 2 or (time.perf_counter() - t_all < budget_s / 2 and len(times) < 10):
            t0 = time.perf_counter()
            cogmen_train_step(model, opt, batch)
            times.append(time.perf_counter() - t0)
        times.sort()
        out[tag] = (n_utt / times[len(times) // 2], len(times))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--max_len", type=int, default=110)
    ap.add_argument("--dataset", default="iemocap-cogmen-sbert-6")  # d_t=768 -> D=1380 as BASELINE.json config 2
    ap.add_argument("--no_graph", action="store_true")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--kernel_reps", type=int, default=200)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world == 1 and args.gpus > 1:
        print("bench.py: --gpus %d needs torch.distributed.run; running 1 rank" % args.gpus, file=sys.stderr)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from erc_amd import capi
    from erc_amd.cogmen import COGMENTrainer
    from erc_amd.engine import GraphedStep, all_reduce_grads
    from erc_amd.params import ERCParams
    capi.lib()  # fail loudly if the HIP library is missing

    params = ERCParams().from_args(["--dataset=" + args.dataset, "--modality=atv", "--compute=" + args.dtype,
                                    "--optim.lr=0.0001", "--optim.weight_decay=1e-8"])
    params.train.batch_size = args.batch
    trainer = COGMENTrainer(params, device)
    host_batch = synthetic_batch(params, args.batch, args.max_len, seed=1 + rank)
    batch = trainer.prepare_batch(host_batch)
    n_utt = int(host_batch["label"].shape[0])

    # ---------------------------------------------------------------- step function
    use_graph = not args.no_graph
    if world == 1:
        step_fn = lambda: trainer.train_step(batch)
        step = GraphedStep(step_fn) if use_graph else step_fn
    else:
        # forward+backward in one graph; the RCCL all-reduce and the optimizer stay eager (2 launches)
        trainer.model.train()
        fb = lambda: trainer.model.loss_and_grads(batch, trainer.class_weight)
        fb_g = GraphedStep(fb) if use_graph else fb

        def step():
            out = fb_g()
            trainer.optim.step(grad_scale=all_reduce_grads(trainer.model.flat))
            return out

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([n_utt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(cnt)
        total_utt = float(cnt.item())
    else:
        total_utt = float(n_utt)
    stats = trainer.model._ws[next(iter(trainer.model._ws))]["stats"].cpu().tolist()

    # ---------------------------------------------------------------- dominant kernel, HIP events on its stream
    roof = None
    if rank == 0:
        roof = trainer.model.dominant_kernel_probe(batch, reps=args.kernel_reps)
        roof = {"bound": "hbm", "kernel": roof["kernel"], "achieved": roof["gbs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": roof["gbs"] / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes": roof["bytes"], "avg_us": roof["us"], "launches_timed": args.kernel_reps}

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = min(16, avail)  # the 1-GPU box's CPU share (16 threads); more threads only slow small ops down
        torch.set_num_threads(cores)
        res = cpu_baseline(params, host_batch)
        cpu = {"value": res["with_dead_encoder"][0], "unit": "utterances/s", "cores": cores, "kind": "port",
               "sample": "%d full train steps of the same B=%d batch (median), oracle/cogmen.py incl. the "
                         "reference's dead Transformer encoder and per-edge python graph build"
                         % (res["with_dead_encoder"][1], args.batch),
               "value_without_dead_encoder": res["without_dead_encoder"][0]}

    if rank == 0:
        value = total_utt * args.steps / elapsed
        line = {
            "metric": "utterances/sec training step, COGMEN IEMOCAP-6 atv", "value": value,
            "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "COGMEN iemocap-cogmen-6 atv train step (BASELINE.json configs[1]): "
                                   "B=%d dialogues/GPU, T=%d, d_a=100 d_t=768 d_v=512 (D=%d), C=%d, "
                                   "graph build + fwd + CE + bwd + Adam" % (args.batch, args.max_len,
                                                                           params.hidden_all, params.n_classes),
                       "utterances_per_step_per_gpu": n_utt, "global_batch_dialogues": args.batch * world,
                       "parallelism": "dp%d" % world, "hip_graph": use_graph,
                       "features_dtype": args.dtype, "loss": stats[0]},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
