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

from . import capi
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
    lo, hi = int(params.get("syn_min_len", lo)), int(params.get("syn_max_len", hi))     # (tests: equal lengths -> repeating shapes)
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


def longest_dialogue(loader):
    """longest training dialogue of this rank's split (the T capacity of the buckets)"""
    if isinstance(loader, StoreLoader):
        return int(loader.store.lengths.max())
    return max(len(d["label"]) for d in loader.dataset.dialogs)


def first_batch(loader):
    """a batch of the loader's shape WITHOUT advancing its shuffling generator (the epochs must see the same permutations
    whether or not a probe was drawn)"""
    if isinstance(loader, StoreLoader):
        return loader.store.batch(torch.arange(min(loader.batch_size, len(loader.store))))
    n = min(loader.batch_size, len(loader.dataset))
    return loader.collate_fn([loader.dataset[i] for i in range(n)])


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
    """Captured HIP graphs of the whole training step, least-recently-used eviction.

    * CAPACITY BUCKETS (trainers that offer ``capacity_bucket``: COGMEN in the bf16 compute mode).  The reference reshuffles
      the dialogues every epoch and its last batch is smaller (lumo/trainer/trainer.py:429-442, mmbase.py:468), so (B, T, N)
      almost never repeats.  A bucket is a set of static capacity-sized input buffers -- ``batch_size`` dialogues (missing
      ones get length 0), the longest dialogue of the split, N rounded up to a multiple of 256 -- plus one graph captured
      over them; the step's kernels read the true node count from the device (COGMENModule.dynamic_n), so every batch
      that fits replays that graph: a reshuffled epoch hits a handful of graphs.
    * otherwise one graph per exact batch shape, captured when the shape shows up the SECOND time (under shuffling most
      shapes never repeat: capturing each one would cost a synchronisation + instantiation per step and pin a workspace).

    Whatever the key, the first step of a key is a real training step run EAGERLY ON THE GRAPH'S OWN STATIC BUFFERS (the
    batch is copied in first): every pointer the later capture sees -- inputs, workspace, weight-gradient table -- already
    exists, so nothing is allocated or uploaded while capturing.  Losses are those of the eager loop."""

    def __init__(self, trainer, maxsize=16, capture=True):
        import collections
        self.trainer, self.maxsize = trainer, maxsize
        self.capture = capture     # False (--graph_capture=False): same buckets and static buffers, every step eager
        self.cache = collections.OrderedDict()     # key -> [static, graph or None, out, workspace, fill, dynamic]
        self.replays = self.eager = self.captures = 0
        self.capture_failed = False
        self.lazy = True           # False (data parallel): only precaptured graphs, everything else eager

    # -- the two places that touch the HIP runtime (a test replaces them with a recorder that, like a real capture, does not
    #    execute what it records)
    def _capture(self, fn):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        return g, out

    def _sync(self):
        torch.cuda.synchronize()

    @staticmethod
    def shape_key(batch):
        return tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(batch.items()) if torch.is_tensor(v)) + \
            tuple((k, v) for k, v in sorted(batch.items()) if isinstance(v, int))

    @staticmethod
    def _copy_in(static, batch):
        for k, v in batch.items():
            if torch.is_tensor(v):
                static[k].copy_(v, non_blocking=True)

    def _run(self, ent, eager):
        static, graph, out, dynamic = ent[0], ent[1], ent[2], ent[5]
        model = self.trainer.model
        if dynamic:
            model.dynamic_n = True
        try:
            if eager:
                return self.trainer.train_step(static)
            graph.replay()
            return out
        finally:
            if dynamic:
                model.dynamic_n = False

    def precapture(self, probe):
        """Data parallel: capture every capacity bucket NOW, in the same order on every rank.  A captured step contains the
        gradient all-reduce; a rank that captured lazily (whenever ITS batch first hit a bucket) would record a collective its
        peers execute.  Each bucket runs one warm-up step on synthetic lengths -- every kernel runs (workspaces, weight-gradient
        tables, the collective) -- and is then captured.  The warm-ups are REAL steps on made-up batches (zero features, label 0;
        the same on every rank): everything they change -- parameters, Adam moments, the optimizer's step count and dropout
        offset, the bf16 weight shadows, BatchNorm's running statistics, the health word -- is snapshotted before and put back
        afterwards, bit for bit, so training starts from the state it was given.  (A raised health word does not suppress the
        update: the step's first launch rolls the word, erc_cogmen_fwd_tile.)"""
        tr = self.trainer
        flat, model = tr.model.flat, tr.model
        opt = getattr(tr, "optim", None)
        saved = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
        snap = [(t, t.clone()) for t in (getattr(flat, n, None) for n in ("data", "exp_avg", "exp_avg_sq", "grad_full")) if t is not None]
        if opt is not None and getattr(opt, "state", None) is not None:
            snap.append((opt.state, opt.state.clone()))
        if getattr(flat, "p2p", None) is not None:
            snap.append((flat.p2p.epoch, flat.p2p.epoch.clone()))
        for key, make, fill, synth in tr.all_capacity_buckets(probe):
            static = make()
            synth(static)
            ent = self.cache[key] = [static, None, None, None, fill, True, 1]
            self._run(ent, eager=True)
            model.dynamic_n = True
            try:
                g, out = self._capture(lambda: tr.train_step(static))
            finally:
                model.dynamic_n = False
            ent[1], ent[2], ent[3] = g, out, getattr(model, "_last_ws", None)
            self.captures += 1
        self._sync()
        with torch.no_grad():
            for t, keep in snap:
                t.copy_(keep)
            sd = model.state_dict()
            for k, v in saved.items():
                sd[k].copy_(v)
        flat.health.zero_()
        flat.events.zero_()
        if getattr(model, "shadows", None) is not None:
            model.refresh_shadows()          # the bf16 copies follow the restored parameters
        self.maxsize = max(self.maxsize, len(self.cache) + 2)

    def step(self, batch, key=None, resident=False):
        """``resident``: ``batch`` lives in fixed device buffers of its own (FixedBatches) -- the graph binds to them."""
        bucket = None
        if key is not None:
            key = (key, )
        else:
            bucket = self.trainer.capacity_bucket(batch) if hasattr(self.trainer, "capacity_bucket") else None
            key = bucket[0] if bucket is not None else self.shape_key(batch)
        ent = self.cache.get(key)
        if ent is None and not self.lazy:
            self.eager += 1
            return self.trainer.train_step(batch)
        if ent is None:
            if bucket is not None:
                static, fill = bucket[1](), bucket[2]
            else:
                static = batch if resident else {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
                fill = None if resident else self._copy_in
            ent = self.cache[key] = [static, None, None, None, fill, bucket is not None]
            ent.append(0)            # occurrences
            while len(self.cache) > self.maxsize:
                self.cache.popitem(last=False)
        else:
            self.cache.move_to_end(key)
        static, graph, fill = ent[0], ent[1], ent[4]
        ent[6] += 1
        if fill is not None and (graph is not None or ent[6] > 1 or ent[5]):
            fill(static, batch)          # (a plain first occurrence was cloned above: already in place)
        if graph is not None:
            self.replays += 1
            return self._run(ent, eager=False)
        stats = self._run(ent, eager=True)                  # a real step, on the static buffers
        self.eager += 1
        if self.capture_failed or not self.capture or not (ent[5] or ent[6] >= 2 or resident):
            return stats                                    # plain shapes are captured on their second occurrence
        try:
            if ent[5]:
                self.trainer.model.dynamic_n = True
            g, out = self._capture(lambda: self.trainer.train_step(static))
        except Exception as exc:                               # a step that cannot be captured stays eager
            print(json.dumps({"graph_replay": "capture failed, staying eager", "error": str(exc)[:200]}), file=sys.stderr)
            self.capture_failed = True
            return stats
        finally:
            if ent[5]:
                self.trainer.model.dynamic_n = False
        # the graph holds raw pointers into this key's workspace: keep the workspace object alive with the graph, whatever
        # the module's own LRU cache does with it (evaluation batches of other shapes come in between)
        ent[1], ent[2], ent[3] = g, out, getattr(self.trainer.model, "_last_ws", None)
        self.captures += 1
        return stats


class ResidentEpochs:
    """``--resident``: the training dialogues live in HBM (datasets.DeviceDialogueStore, 288 GB per GPU: IEMOCAP's features are
    16 MB) and a step's batch is never materialised.  Per epoch the host draws the permutation (DataLoader(shuffle=True)
    semantics: every dialogue once, a smaller last batch) and uploads ONE int32 table [steps, 2 B] (lengths | first store
    rows of every batch's dialogues); per step it copies that step's 2 B int32 into the fixed descriptor buffer and replays
    the bucket's captured graph -- the projection launch reads feature rows, speakers and labels straight from the store
    (csrc/cogmen_project.hip, resident mode).  Host work per step: one 256-byte device copy + one graph launch."""

    N_BUCKET = 128

    def __init__(self, trainer, store, batch_size, seed, capture=True):
        self.trainer, self.store, self.B = trainer, store, int(batch_size)
        self.gen = torch.Generator().manual_seed(seed)
        self.T = int(store.lengths.max())
        dev = store.device
        self.cur_desc = torch.zeros(2 * self.B, dtype=torch.int32, device=dev)
        self.acc = torch.zeros(4, dtype=torch.float64, device=dev)      # sums of the steps' {loss, #correct, weight, -}
        self.graphs, self.capture = {}, capture
        self.replays = self.eager = self.captures = 0
        self._lens32, self._offs32 = store.lengths.to(torch.int32), store.offsets[:-1].to(torch.int32)
        self._ahead = []           # (node counts per step, device table) of the epochs planned ahead

    def plan(self, n_epochs):
        """Draw the permutations of the next ``n_epochs`` epochs (sequential randperm draws: the same epochs as drawing them
        one by one) and upload their batch tables -- int32 [steps, 2 B]: lengths | first store rows, 0 for the empty slots of
        a smaller last batch -- in one copy: nothing but the step loop is left inside an epoch."""
        import numpy as np
        st, B = self.store, self.B
        n = len(st)
        steps = -(-n // B)
        lens, offs = self._lens32.numpy(), self._offs32.numpy()
        tabs = np.zeros((n_epochs, steps, 2 * B), dtype=np.int32)
        for e in range(n_epochs):
            order = torch.randperm(n, generator=self.gen).numpy()
            flat_l, flat_o = np.zeros(steps * B, dtype=np.int32), np.zeros(steps * B, dtype=np.int32)
            flat_l[:n], flat_o[:n] = lens[order], offs[order]
            tabs[e, :, :B], tabs[e, :, B:] = flat_l.reshape(steps, B), flat_o.reshape(steps, B)
        dev = torch.from_numpy(tabs).to(st.device)
        for e in range(n_epochs):
            self._ahead.append((tabs[e, :, :B].sum(1).tolist(), dev[e]))

    def supported(self):
        """Can EVERY step of an epoch run from the resident store?  Probed with the smallest bucket and with the largest a batch
        of this store can need (batch_size of its longest dialogues: above the fused path's node limit `resident_batch` returns
        None -- found here, before the first epoch, not by a step in the middle of one)."""
        lens = sorted((int(v) for v in self.store.lengths.tolist()), reverse=True)
        worst = min(-(-sum(lens[:self.B]) // self.N_BUCKET) * self.N_BUCKET, self.B * self.T)
        return all(self.trainer.resident_batch(self.store, self.cur_desc, self.B, self.T, cap) is not None
                   for cap in sorted({self.N_BUCKET, max(worst, self.N_BUCKET)}))

    def _step_fn(self, batch):
        stats = self.trainer.train_step(batch)
        self.acc.add_(stats[:4])
        return stats

    def epoch(self):
        """one pass over the store; returns (#utterances, #steps)"""
        st, B = self.store, self.B
        n = len(st)
        steps = -(-n // B)
        if not self._ahead:
            self.plan(1)
        counts, table_dev = self._ahead.pop(0)
        model = self.trainer.model
        model.dynamic_n = True
        try:
            for s in range(steps):
                self.cur_desc.copy_(table_dev[s], non_blocking=True)
                n_cap = min(-(-counts[s] // self.N_BUCKET) * self.N_BUCKET, B * self.T)
                ent = self.graphs.get(n_cap)
                if ent is not None and ent[0] is not None:
                    ent[0].replay()
                    self.replays += 1
                    continue
                batch = self.trainer.resident_batch(st, self.cur_desc, B, self.T, n_cap)
                self._step_fn(batch)                    # a real step; allocates the bucket's workspace
                self.eager += 1
                if not self.capture:
                    continue
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._step_fn(batch)
                self.graphs[n_cap] = (g, batch, getattr(model, "_last_ws", None))
                self.captures += 1
        finally:
            model.dynamic_n = False
        return sum(counts), steps


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
    # captured whole-step graphs (StepGraphs): capacity buckets where the trainer offers them, exact shapes otherwise.  Under
    # data parallelism the captured step is the same one -- forward, backward, the RCCL all-reduce, the optimizer -- but
    # every capture must happen at the same point on every rank: buckets are captured up front, in one order
    # (StepGraphs.precapture); a trainer without buckets keeps the eager step there.
    if hasattr(trainer, "capacity_bucket"):
        trainer.t_cap = longest_dialogue(train_loader)
    graphs = StepGraphs(trainer, capture=params.get("graph_capture", True)) if params.get("graph_replay", True) else None
    fixed = FixedBatches(train_loader, trainer, params.seed + rank) if params.get("fixed_batches", False) else None
    if graphs is not None and world > 1 and fixed is None:
        probe = trainer.prepare_batch(first_batch(train_loader))
        if hasattr(trainer, "all_capacity_buckets") and trainer.all_capacity_buckets(probe):
            graphs.precapture(probe)
            graphs.lazy = False         # a batch outside every bucket runs eagerly (same collectives, no capture)
        else:
            graphs = None
    n_steps = len(fixed) if fixed is not None else len(train_loader)
    ring = torch.zeros(max(1, n_steps), 4, dtype=torch.float32, device=device)    # per-step {loss, #correct, ...}: read once per epoch
    resident, acc_prev = None, [0.0] * 4
    if params.get("resident", False):
        if world > 1 or not isinstance(train_loader, StoreLoader) or not hasattr(trainer, "resident_batch"):
            raise SystemExit("--resident needs --device_collate, one rank and a trainer with resident batches (cogmen, --compute=bf16)")
        resident = ResidentEpochs(trainer, train_loader.store, params.train.batch_size, params.seed + rank,
                                  capture=params.get("graph_capture", True))
        if not resident.supported():
            raise SystemExit("--resident: this configuration cannot run the fused bf16 step in capacity mode")
        resident.plan(params.epoch)
    for epoch in range(params.epoch):
        trainer.model.train()
        t0, n_utt, counts = time.perf_counter(), 0, []
        if resident is not None:
            n_utt, n_st = resident.epoch()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tot = resident.acc.cpu().tolist()                           # running totals: this epoch = the difference
            acc, acc_prev = [a - b for a, b in zip(tot, acc_prev)], tot
            if rank == 0 and params.log_every:
                print(json.dumps({"epoch": epoch, "steps": n_st, "Lall": acc[0] / n_st, "Acc": acc[1] / max(1, n_utt)}), flush=True)
        for i, item in enumerate(() if resident is not None else (fixed if fixed is not None else train_loader)):
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
        if resident is None:
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
                              "best": best, "graph_replays": (resident or graphs).replays if (resident or graphs) else 0,
                              "eager_steps": (resident or graphs).eager if (resident or graphs) else len(counts),
                              "graphs_captured": (resident or graphs).captures if (resident or graphs) else 0}), flush=True)
    if params.get("save") and rank == 0:
        from . import checkpoint
        checkpoint.save(trainer, params.save)
    if world > 1:
        # the captured step graphs hold RCCL work: they go before the process group does (bench.py saw an abort at interpreter
        # exit, once in a while, with the group torn down under live graphs)
        graphs = resident = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return best
