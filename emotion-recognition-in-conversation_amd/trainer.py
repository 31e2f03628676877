"""Minimal run driver behind the reference's plugin surface.

Replaces what ``mmbase.main`` gets from lumo (track_mm/mmbase.py:483-499,
lumo/trainer/trainer.py:402-442): build params from the command line, build the
data loaders (``ERCCollate`` batch layout), loop ``epoch`` times over
``train_step``, evaluate with ``test_step`` after every epoch
(EvalCallback(test_per_epoch=1), mmbase.py:136) and report the sklearn metric
set of mmbase.py:253-323.  One process per GPU; under torch.distributed the
trainers all-reduce the flat gradient buffer (RCCL) once per step.

There are no dataset pickles offline: ``--synthetic`` (default) draws seeded
IEMOCAP-/MELD-shaped dialogues (synthetic.py).
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


class ListDataset(torch.utils.data.Dataset):
    """Yields 1-element lists like lumo's DatasetBuilder (lumo/data/builder.py:100-101)."""

    def __init__(self, dialogs):
        self.dialogs = dialogs

    def __len__(self):
        return len(self.dialogs)

    def __getitem__(self, i):
        return [self.dialogs[i]]


def make_loaders(params, rank=0, world=1):
    if not params.get("synthetic", True):
        raise NotImplementedError("real-data pickle readers are a 'next' row (SURVEY.md 8f-2); use --synthetic")
    meld = "meld" in params.dataset
    lo, hi = (1, 33) if meld else (20, 110)
    mk = lambda n, seed: make_dialogues(n, params.dims(), n_speakers=params.n_speakers, n_classes=params.n_classes,
                                        min_len=lo, max_len=hi, seed=seed)
    train = ListDataset(mk(params.n_train, params.seed + 1000 * rank))   # every rank draws its own batches
    test = ListDataset(mk(params.n_test, params.seed + 7))
    collate = ERCCollate(params)
    gen = torch.Generator().manual_seed(params.seed + rank)
    tl = DataLoader(train, batch_size=params.train.batch_size, shuffle=True, collate_fn=collate,
                    num_workers=params.train.get("num_workers", 0) or 0, generator=gen)
    el = DataLoader(test, batch_size=params.test.batch_size, shuffle=False, collate_fn=collate, num_workers=0)
    return tl, el


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
    trainer = trainer_cls(params, device)
    train_loader, test_loader = make_loaders(params, rank, world)
    best = {}
    for epoch in range(params.epoch):
        trainer.model.train()
        t0, n_utt, last = time.perf_counter(), 0, None
        for i, batch in enumerate(train_loader):
            dev_batch = trainer.prepare_batch(batch)
            stats = trainer.train_step(dev_batch)
            n_utt += int(batch["label"].shape[0])
            if rank == 0 and params.log_every and (i + 1) % params.log_every == 0:
                s = stats.tolist()  # one device->host sync per logged step
                last = {"epoch": epoch, "step": i, "Lall": s[0], "Acc": s[1] / max(1, batch["label"].shape[0])}
                print(json.dumps(last), flush=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # test after every epoch (mmbase.py:136,180-201)
        trainer.model.eval()
        true, pred = [], []
        for batch in test_loader:
            logits = trainer.to_logits(trainer.prepare_batch(batch))
            if logits.dim() == 3:
                logits = logits[batch["attention_mask"].bool().to(logits.device)]
            pred.extend(logits.argmax(-1).cpu().tolist())
            true.extend(batch["label"].tolist())
        if rank == 0:
            rep = classification_report(true, pred, params.n_classes)
            for k in ("acc", "wa", "f1", "mif1", "maf1", "pre", "rec"):
                best[k] = max(best.get(k, 0.0), rep[k])
            print(json.dumps({"epoch": epoch, "train_utt_per_s": n_utt / dt, "test": {k: rep[k] for k in rep if k != "cm"},
                              "best": best}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    return best
