"""Minimal run driver behind the reference's plugin surface.

Replaces what ``mmbase.main`` gets from lumo (track_mm/mmbase.py:483-499,
lumo/trainer/trainer.py:402-442): build params from the command line, build the
data loaders (``ERCCollate`` batch layout), loop ``epoch`` times over
``train_step``, evaluate with ``test_step`` after every epoch
(EvalCallback(test_per_epoch=1), mmbase.py:136) and report the sklearn metric
set of mmbase.py:253-323.  One process per GPU; under torch.distributed the
trainers all-reduce the flat gradient buffer (RCCL) once per step.

There are no dataset pickles offline: ``--synthetic`` (default) draws seeded
IEMOCAP-/MELD-shaped dialogues (synthetic.py); ``--synthetic=False --data_root=<dir>``
(or $ERC_IEMOCAP_ROOT / $ERC_MELD_ROOT) reads the reference's feature pickles
(datasets.py).  ``--device_collate`` keeps the dialogues resident in HBM and builds
every batch on the device (datasets.DeviceDialogueStore) instead of in a DataLoader.
"""
import json
import os
import sys
import time

import numpy as np
import torch
from torch.utils.data import DataLoader

from .collate import ERCCollate
from .synthetic import make_dialogues


def fix_seed(seed):
    """What ``trainer.rnd.mark(seed)`` amounts to on a first run (lumo/trainer/rnd.py:17-33, lumo/utils/random.py:22-43):
    python / numpy / torch generators seeded, so parameter initialisation and shuffling repeat run to run."""
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class ListDataset(torch.utils.data.Dataset):
    """Yields 1-element lists like lumo's DatasetBuilder (lumo/data/builder.py:100-101)."""

    def __init__(self, dialogs):
        self.dialogs = dialogs

    def __len__(self):
        return len(self.dialogs)

    def __getitem__(self, i):
        return [self.dialogs[i]]


class StoreLoader:
    """DataLoader-shaped iterator over a DeviceDialogueStore: seeded shuffle on the host, batch assembly on the device."""

    def __init__(self, store, batch_size, shuffle, seed):
        self.store, self.batch_size, self.shuffle = store, batch_size, shuffle
        self.gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return -(-len(self.store) // self.batch_size)

    def __iter__(self):
        n = len(self.store)
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        for i in range(0, n, self.batch_size):
            yield self.store.batch(order[i:i + self.batch_size])


def shard_dialogues(dialogs, rank, world):
    """This rank's share of the training dialogues, EQUAL in size on every rank: the list is padded to a multiple of
    ``world`` by wrapping around (what torch's DistributedSampler -- the sampler accelerate gives the reference,
    lumo/trainer/trainer.py:377-384 -- does with drop_last=False), then dealt round-robin.  Equal shard sizes mean an
    equal number of batches, hence an equal number of gradient all-reduces per epoch on every rank: with plain
    ``dialogs[rank::world]`` 259 dialogues over 2 ranks at batch 43 gave 4 vs 3 steps and the rank with the extra
    batch waited in its all-reduce forever."""
    n = len(dialogs)
    if world <= 1 or n == 0:
        return list(dialogs)
    total = -(-n // world) * world
    idx = [i % n for i in range(total)]
    return [dialogs[i] for i in idx[rank::world]]


def load_dialogues(params, rank=0, world=1):
    """(train dialogues of this rank, test dialogues)."""
    if not params.get("synthetic", True):
        from .datasets import read_dialogues
        roots = params.get("data_root", None)
        train = read_dialogues(params.dataset, "train", roots)
        return shard_dialogues(train, rank, world), read_dialogues(params.dataset, "test", roots)
    meld = "meld" in params.dataset
    lo, hi = (1, 33) if meld else (20, 110)
    mk = lambda n, seed: make_dialogues(n, params.dims(), n_speakers=params.n_speakers, n_classes=params.n_classes,
                                        min_len=lo, max_len=hi, seed=seed)
    return mk(params.n_train, params.seed + 1000 * rank), mk(params.n_test, params.seed + 7)   # every rank draws its own


def make_loaders(params, rank=0, world=1, device=None):
    train_d, test_d = load_dialogues(params, rank, world)
    if params.get("device_collate", False):
        from .datasets import DeviceDialogueStore
        dt = torch.bfloat16 if params.compute == "bf16" else torch.float32
        return (StoreLoader(DeviceDialogueStore(train_d, params, device, dt), params.train.batch_size, True, params.seed + rank),
                StoreLoader(DeviceDialogueStore(test_d, params, device, dt), params.test.batch_size, False, 0))
    train, test = ListDataset(train_d), ListDataset(test_d)
    collate = ERCCollate(params)
    gen = torch.Generator().manual_seed(params.seed + rank)
    tl = DataLoader(train, batch_size=params.train.batch_size, shuffle=True, collate_fn=collate,
                    num_workers=params.train.get("num_workers", 0) or 0, generator=gen)
    el = DataLoader(test, batch_size=params.test.batch_size, shuffle=False, collate_fn=collate, num_workers=0)
    return tl, el


class FixedBatches:
    """``--fixed_batches``: the training batches are collated once and kept on the device; an epoch visits them in a
    freshly shuffled ORDER.  Batch shapes then repeat every epoch, so every step after the first epoch is one HIP-graph
    replay with no host-side collate and no host-to-device copy.  (Opt-in: the reference reshuffles the dialogues
    themselves every epoch.)"""

    def __init__(self, loader, trainer, seed):
        self.batches = [(int(b["label"].shape[0]), trainer.prepare_batch(b)) for b in loader]
        self.gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for i in torch.randperm(len(self.batches), generator=self.gen).tolist():
            yield i, self.batches[i]


class StepGraphs:
    """One captured HIP graph of the whole training step per batch shape (B, T, N), least-recently-used eviction.  The
    FIRST step of a shape runs eagerly -- it is a real training step and allocates that shape's workspace -- and is then
    captured (capture records, it does not execute); later steps of the shape copy the batch into the graph's static
    input buffers and replay.  Losses are therefore identical to the eager loop."""

    def __init__(self, trainer, maxsize=16):
        import collections
        self.trainer, self.maxsize = trainer, maxsize
        self.cache = collections.OrderedDict()
        self.replays = self.eager = 0

    @staticmethod
    def shape_key(batch):
        return tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(batch.items()) if torch.is_tensor(v)) + \
            tuple((k, v) for k, v in sorted(batch.items()) if isinstance(v, int))

    def step(self, batch, key=None, resident=False):
        """``resident``: ``batch`` lives in fixed device buffers of its own (FixedBatches) -- the graph binds to them."""
        key = (key, ) if key is not None else self.shape_key(batch)
        ent = self.cache.get(key)
        if ent is not None:
            static, graph, out = ent[:3]
            self.cache.move_to_end(key)
            if not resident:
                for k, v in batch.items():
                    if torch.is_tensor(v):
                        static[k].copy_(v, non_blocking=True)
            graph.replay()
            self.replays += 1
            return out
        stats = self.trainer.train_step(batch)                # the real step of this shape's first occurrence
        self.eager += 1
        static = batch if resident else {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        try:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.trainer.train_step(static)
        except Exception as exc:                               # a step that cannot be captured stays eager
            print(json.dumps({"graph_replay": "capture failed, staying eager", "error": str(exc)[:200]}), file=sys.stderr)
            self.step = lambda b, key=None, resident=False: self.trainer.train_step(b)
            return stats
        # the graph holds raw pointers into this shape's workspace: keep the workspace object alive with the graph, whatever
        # the module's own LRU cache does with it (evaluation batches of other shapes come in between)
        self.cache[key] = (static, graph, out, getattr(self.trainer.model, "_last_ws", None))
        while len(self.cache) > self.maxsize:
            self.cache.popitem(last=False)
        return stats


def classification_report(true, pred, n_classes):
    """The metric set of mmbase.py:259-275."""
    from sklearn import metrics
    return {
        "cm": metrics.confusion_matrix(true, pred, labels=range(n_classes)).tolist(),
        "acc": metrics.accuracy_score(true, pred),
        "wa": metrics.balanced_accuracy_score(true, pred),
        "pre": metrics.precision_score(true, pred, average="weighted", zero_division=0),
        "rec": metrics.recall_score(true, pred, average="weighted", zero_division=0),
        "f1": metrics.f1_score(true, pred, average="weighted", zero_division=0),
        "mif1": metrics.f1_score(true, pred, average="micro", zero_division=0),
        "maf1": metrics.f1_score(true, pred, average="macro", zero_division=0),
    }


def run(trainer_cls, params_cls, argv=None):
    params = params_cls()
    params.from_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("the ERC hot path runs on an MI355X through libercgraft.so; no GPU is visible "
                           "(there is no CPU fallback)")
    device = torch.device(params.device if params.device not in (None, "cuda") else "cuda:%d" % local)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    fix_seed(params.seed)
    trainer = trainer_cls(params, device)
    if params.get("load"):      # checkpoint in the reference's envelope (checkpoint.py)
        from . import checkpoint
        checkpoint.load(trainer, params.load)
    train_loader, test_loader = make_loaders(params, rank, world, device)
    best = {}
    # one captured graph per batch shape; with gradient exchange (world > 1) the step stays eager: the all-reduce is
    # issued from the host between backward and optimizer
    graphs = StepGraphs(trainer) if (params.get("graph_replay", True) and world == 1) else None
    fixed = FixedBatches(train_loader, trainer, params.seed + rank) if params.get("fixed_batches", False) else None
    n_steps = len(fixed) if fixed is not None else len(train_loader)
    ring = torch.zeros(max(1, n_steps), 4, dtype=torch.float32, device=device)    # per-step {loss, #correct, ...}: read once per epoch
    for epoch in range(params.epoch):
        trainer.model.train()
        t0, n_utt, counts = time.perf_counter(), 0, []
        for i, item in enumerate(fixed if fixed is not None else train_loader):
            if fixed is not None:
                bid, (n_b, dev_batch) = item
            else:
                bid, n_b, dev_batch = None, int(item["label"].shape[0]), trainer.prepare_batch(item)
            stats = graphs.step(dev_batch, key=bid, resident=fixed is not None) if graphs is not None else trainer.train_step(dev_batch)
            ring[i].copy_(stats[:4], non_blocking=True)          # no device->host synchronisation inside the epoch
            n_utt += n_b
            counts.append(n_b)
        torch.cuda.synchronize()
        if hasattr(trainer.model, "check_cluster"):
            trainer.model.check_cluster()
        dt = time.perf_counter() - t0
        if rank == 0 and params.log_every:
            rows = ring[:len(counts)].cpu().tolist()
            for i, (row, n_b) in enumerate(zip(rows, counts)):
                if (i + 1) % params.log_every == 0:
                    print(json.dumps({"epoch": epoch, "step": i, "Lall": row[0], "Acc": row[1] / max(1, n_b)}), flush=True)
        # test after every epoch (mmbase.py:136,180-201)
        trainer.model.eval()
        true, pred = [], []
        for batch in test_loader:
            logits = trainer.to_logits(trainer.prepare_batch(batch))
            if logits.dim() == 3:
                logits = logits[batch["attention_mask"].bool().to(logits.device)]
            pred.extend(logits.argmax(-1).cpu().tolist())
            true.extend(batch["label"].tolist())
        if hasattr(trainer.model, "check_cluster"):
            trainer.model.check_cluster()      # a timeout inside to_logits would make the metrics below meaningless
        if rank == 0:
            rep = classification_report(true, pred, params.n_classes)
            for k in ("acc", "wa", "f1", "mif1", "maf1", "pre", "rec"):
                best[k] = max(best.get(k, 0.0), rep[k])
            print(json.dumps({"epoch": epoch, "train_utt_per_s": n_utt / dt, "test": {k: rep[k] for k in rep if k != "cm"},
                              "best": best, "graph_replays": graphs.replays if graphs else 0,
                              "eager_steps": graphs.eager if graphs else len(counts)}), flush=True)
    if params.get("save") and rank == 0:
        from . import checkpoint
        checkpoint.save(trainer, params.save)
    if world > 1:
        torch.distributed.destroy_process_group()
    return best
