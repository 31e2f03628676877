"""Host-side plumbing shared by the four ERC plugins.

* ``FlatParams``  : the LIVE parameters of a module packed into one fp32 device
  buffer (plus flat grad / Adam moment buffers of the same layout).  The
  ``nn.Parameter``s of the module become views into it, so ``state_dict`` keys
  and shapes stay those of the reference while the optimizer and the RCCL
  gradient all-reduce see one contiguous array.  Parameters that never receive
  a gradient in the reference (dead encoder, unused heads) are
  left out: torch.optim skips ``grad is None`` parameters, so they never change.
* ``GemmPlanner`` : split-K policy + slab workspace + the batched slab-reduce
  job table for weight gradients.
* ``FusedAdam``   : one erc_adam_step over the flat buffer (optionally after a
  gradient all-reduce over RCCL).
"""
import math
import os

import torch

from . import capi

ALIGN = 64  # floats: every group starts on a 256-byte boundary


class WorkspaceCache:
    """Per-module workspace sets keyed by batch shape, least-recently-used eviction.  In the real training loop
    (shuffled batches, smaller last batch: mmbase.py:468) the number of valid utterances N changes almost every step;
    an unbounded ``dict`` of (B, T, N) -> workspace would grow by hundreds of MB per new shape (DAG-ERC: ~350 MB).
    ``maxsize`` shapes stay resident (env ERC_WS_CACHE).  A HIP graph captured over a workspace keeps the workspace OBJECT
    alive itself (trainer.StepGraphs stores it next to the graph), so eviction here never frees memory a graph points to."""

    def __init__(self, maxsize=None):
        import collections
        self.maxsize = int(os.environ.get("ERC_WS_CACHE", 4)) if maxsize is None else maxsize
        self._d = collections.OrderedDict()
        self.last = None

    def get(self, key, make):
        ws = self._d.get(key)
        if ws is None:
            ws = make()
            self._d[key] = ws
            while len(self._d) > self.maxsize:
                victim = next((k for k in self._d if k != key), None)
                if victim is None:
                    break
                del self._d[victim]
        else:
            self._d.move_to_end(key)
        self.last = ws
        return ws

    def values(self):
        return list(self._d.values())

    def items(self):
        return list(self._d.items())

    def __len__(self):
        return len(self._d)


class FlatParams:
    def __init__(self, groups, device):
        """``groups``: list of lists of (name, nn.Parameter).  Members of a group are
        laid out back to back (so e.g. q|k|v|skip weights form one [4F,F] matrix)."""
        self.device = torch.device(device)
        self.offsets, self.shapes = {}, {}
        off = 0
        for grp in groups:
            off = (off + ALIGN - 1) // ALIGN * ALIGN
            for name, p in grp:
                self.offsets[name] = off
                self.shapes[name] = tuple(p.shape)
                off += p.numel()
        self.numel = (off + ALIGN - 1) // ALIGN * ALIGN
        self.data = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        # the gradient buffer carries one extra group: its first word is the step's HEALTH WORD (ercgraft.h, erc_health_roll),
        # raised by a persistent kernel whose bounded poll timed out.  Living behind the gradients, it travels in the step's
        # one all-reduce, so under data parallelism every rank skips the update when any rank raised it.
        self.grad_full = torch.zeros(self.numel + ALIGN, dtype=torch.float32, device=self.device)
        self.grad = self.grad_full[:self.numel]
        self.health = self.grad_full[self.numel:self.numel + 1].view(torch.int32)
        self.events = torch.zeros(2, dtype=torch.int32, device=self.device)   # [0]: steps skipped since the last check_health
        self.exp_avg = torch.zeros_like(self.data)
        self.exp_avg_sq = torch.zeros_like(self.data)
        self.params = {}
        with torch.no_grad():
            for grp in groups:
                for name, p in grp:
                    view = self.view(self.data, name)
                    view.copy_(p.detach().to(self.device, torch.float32))
                    p.data = view
                    p.grad = self.view(self.grad, name)
                    self.params[name] = p
        self.live_numel = sum(p.numel() for p in self.params.values())

    def view(self, flat, name):
        off = self.offsets[name]
        n = 1
        for s in self.shapes[name]:
            n *= s
        return flat[off:off + n].view(self.shapes[name])

    def w(self, name):
        return self.view(self.data, name)

    def roll_health(self):
        """Start of a training step: a health word the previous step left raised becomes one counted event and is
        cleared (one launch, no host synchronisation)."""
        capi.health_roll(self.health, self.events)

    def check_health(self, what, partial=False):
        """Host-side report (one device->host copy; the trainer calls it once per epoch): raises if any step since the
        last call was skipped on the device, or if the word is raised right now (evaluation pass).
        ``partial``: the module's timeouts come from the optimizer INSIDE the weight-gradient launch or from the peer-to-peer
        exchange, which give up per gradient tile / chunk -- the parameters may then be a mix of updated and skipped tiles:
        the buffer is marked ``tainted`` (checkpoint.save refuses, the planner keeps the two-launch form from here on)."""
        ev, live = int(self.events[0].item()), int(self.health[0].item())
        if ev or live:
            self.events.zero_()
            self.health.zero_()
            n = ev + (1 if live else 0)
            if partial:
                self.tainted = True
                raise capi.ErcGraftError("%s: a bounded wait between cooperating workgroups timed out in %d step(s) (not all of "
                                         "them were resident at once, e.g. another process holds CUs).  This launch gives up per "
                                         "gradient tile / chunk: tiles whose wait timed out kept their old parameters while the "
                                         "others were updated%s.  Treat the run as failed and restart from the last checkpoint "
                                         "(ERC_FUSE_ADAM=0 / ERC_DP_P2P=0 select the optimizer launch that skips a step as a whole)."
                                         % (what, n, ", and ranks may have diverged" if _world_size() > 1 else ""))
            raise capi.ErcGraftError("%s: a bounded wait between cooperating workgroups timed out (not all of them were "
                                     "resident at once, e.g. another process holds CUs); %d optimizer step(s) were skipped "
                                     "on the device%s" % (what, n, " on every rank" if _world_size() > 1 else ""))

    def g(self, name):
        return self.view(self.grad, name)


def _world_size():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class _Tagged:
    """view of a workspace dict whose keys carry a suffix"""

    def __init__(self, d, tag):
        self.d, self.tag = d, tag

    def get(self, k, default=None):
        return self.d.get(k + self.tag, default)

    def __getitem__(self, k):
        return self.d[k + self.tag]

    def __setitem__(self, k, v):
        self.d[k + self.tag] = v


class GemmPlanner:
    """Chooses split-K per GEMM and hands out slab space from one workspace."""

    TARGET_WG = 512
    MIN_CHUNKS = 8
    BK = 32
    MAX_SPLIT = 64

    def __init__(self, device, ws_floats, grad=None):
        self.device = device
        self.ws = torch.empty(ws_floats, dtype=torch.float32, device=device)
        self.grad = grad      # flat gradient buffer: unsplit weight gradients are written there directly
        self.reset()

    def reset(self):
        self.cursor = 0
        self.jobs = []
        self.max_numel = 0
        self.deferred = []     # (A, lda, B, ldb, C, ldc, M, N, K, ones, bias_out): one batched launch at the end
        self.flushed = []      # ... the records of this step that were launched already (bench.py counts their FLOPs)
        self.deferred16 = []   # bf16 compute mode: records of the bf16 weight-gradient launch (defer16)
        self.ranges16 = []     # ... and finished gradient ranges its fused optimizer has to cover (defer16_range)
        self.adam_fused = False

    # deferred weight gradients: 0 / False = exact fp32 matrix cores; 1 / True = bf16 matrix cores, operands rounded (COGMEN
    # bf16 compute mode); 2 = three-term bf16 split of fp32 operands (fp32-class results, ~2x the fp32 instruction's rate)
    mma_bf16 = 0

    def defer(self, A, lda, B, ldb, Cm, ldc, M, N, K, ones, bias_out, gather=None, scale=1.0, mma_bf16=None):
        """C[M,N] = scale * A[K,M]^T B[gather(K),N] (+ bias strip); B may be the bf16 feature block."""
        mb = self.mma_bf16 if mma_bf16 is None else mma_bf16
        self.deferred.append((A, lda, B, ldb, Cm, ldc, M, N, K, ones, bias_out, gather, float(scale), int(mb)))

    WG_STEPS = 64   # k-steps (4 k each) per work item: 16 per wavefront (measured best of 48..128 on COGMEN B=32)

    def flush_wgrads(self, cache, tag=""):
        """Run every weight-gradient product deferred so far as ONE launch (erc_wgrad_table, csrc/wgrad.hip) and forget them.
        The descriptor table, the partial-tile slabs and the per-tile arrival counters are built on the first call
        and reused (operands live in fixed workspace buffers); ``tag`` keeps the tables of a step's several flushes apart
        (an early flush on a side stream: DGCNModule.loss_and_grads)."""
        if not self.deferred:
            return
        import struct
        deferred, self.deferred = self.deferred, []
        self.flushed += deferred
        cache = _Tagged(cache, tag)
        key = tuple((a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, g.data_ptr() if g is not None else 0, sc, mb)
                    for a, _, b, _, c, _, M, N, K, _, _, g, sc, mb in deferred)
        if cache.get("wgrad_key") != key:
            cap = capi.wgrad_max_k_per_split()
            raw, items, tiles, bases = [], 0, 0, []
            steps = int(os.environ.get("ERC_WG_STEPS", 0))
            max_split = int(os.environ.get("ERC_WG_MAXSPLIT", 32))
            if steps <= 0:
                # 64 k-steps per item (16 per wavefront) while all items of the launch fit the chip at once (768 workgroup
                # slots: COGMEN, DialogueGCN); with several rounds of items the per-item overhead (LDS reduce, slab, arrival)
                # is paid per round, and longer items win (measured: MMGCN 4.68 -> 4.47 ms, DAG-ERC 3.80 -> 3.71 ms at 256)
                n64 = sum(-(-M // 64) * -(-N // 64) * max(1, min(32, (-(-K // 4) + 32) // 64))
                          for _, _, _, _, _, _, M, N, K, _, _, _, _, _ in deferred)
                # (round 4, one launch for all of MMGCN's records: 128 / 256 / 512 / 1024 steps -> 3.45 / 3.41 / 3.35 / 3.35 ms;
                #  DAG-ERC 256 / 512 / 1024 -> 3.55 / 3.52 / 3.50 ms)
                steps = self.WG_STEPS if n64 <= 1152 else 512
            for a, lda, b, ldb, c, ldc, M, N, K, ones, bo, g, sc, mb in deferred:
                bf16 = b.dtype == torch.bfloat16
                a_bf16 = a.dtype == torch.bfloat16
                if (not a_bf16 and a.dtype != torch.float32) or c.dtype != torch.float32 or \
                        (not bf16 and b.dtype != torch.float32) or (a_bf16 and bf16):
                    raise capi.ErcGraftError("wgrad table: operand dtypes %s %s %s" % (a.dtype, b.dtype, c.dtype))
                nks = -(-K // 4)
                splits = max(1, min(max_split, (nks + steps // 2) // steps))
                splits = max(splits, -(-K // cap))
                per = -(-nks // splits)
                splits = -(-nks // per)                       # no empty split
                if per * 4 > cap:
                    raise capi.ErcGraftError("wgrad table: K=%d needs more than %d splits" % (K, max_split))
                tm, tn = -(-M // 64), -(-N // 64)
                vec = (1 if (M % 4 == 0 and lda % 4 == 0 and a.data_ptr() % (8 if a_bf16 else 16) == 0) else 0) \
                    | (2 if (N % 4 == 0 and ldb % 4 == 0 and b.data_ptr() % (8 if bf16 else 16) == 0) else 0) \
                    | (4 if (N % 4 == 0 and ldc % 4 == 0 and c.data_ptr() % 16 == 0) else 0)
                n_it = tm * tn * splits
                if mb and (vec & 3) != 3:
                    mb = 0          # the bf16 matrix-core paths need vector access to both operands
                if mb == 2 and (bf16 or a_bf16):
                    mb = 0          # the three-term split is for fp32 operands
                raw.append(struct.pack("<QQQQQ14ifii4x", a.data_ptr(), b.data_ptr(), c.data_ptr(),
                                       bo.data_ptr() if bo is not None else 0, g.data_ptr() if g is not None else 0,
                                       lda, ldb, ldc, M, N, K, ones if bo is not None else 0, int(bf16), splits, tn,
                                       items, n_it, tiles, vec, sc, int(a_bf16), int(mb)))
                bases.append(items)
                items += n_it
                tiles += tm * tn
            cache["wgrad_table"] = torch.frombuffer(bytearray(b"".join(raw)), dtype=torch.uint8).to(self.device)
            cache["wgrad_slabs"] = torch.empty(items * capi.wgrad_slab_floats(), dtype=torch.float32, device=self.device)
            cache["wgrad_counters"] = torch.zeros(tiles, dtype=torch.int32, device=self.device)
            cache["wgrad_items"] = items
            import ctypes
            cache["wgrad_bases"] = (ctypes.c_int32 * len(bases))(*bases)
            cache["wgrad_key"] = key
            cache["wgrad_x3"] = any(d[13] == 2 for d in deferred)
        (capi.wgrad_table_x3 if cache["wgrad_x3"] else capi.wgrad_table)(
            cache["wgrad_table"], len(deferred), cache["wgrad_bases"], cache["wgrad_items"], cache["wgrad_slabs"],
            cache["wgrad_counters"])

    # ------------------------------------------------------------------ bf16 weight gradients (csrc/wgrad_bf16.hip)
    def defer16(self, A, lda, B, ldb, Cm, ldc, M, N, K, ct=False, bias_a=None, bias_b=None, gather=None, k_dev=None):
        """C[M,N] = A[K,M]^T B[gather(K),N] with both operands bf16 in memory (COGMEN bf16 mode); ``ct`` stores C
        transposed (C[n * ldc + m]); bias_a / bias_b receive the fp32 column sums of A / B; ``k_dev`` (device int32): K is a
        capacity, the true row count is read on the device."""
        if self.split_terms > 1:     # split compute modes: fp32 operands, expanded into bf16 terms in registers (erc_wgrad_split)
            if A.dtype != torch.float32 or B.dtype != torch.float32 or Cm.dtype != torch.float32:
                raise capi.ErcGraftError("split wgrad: operand dtypes %s %s %s" % (A.dtype, B.dtype, Cm.dtype))
            if M > 128 or M % 4 or lda % 4 or lda < M or ldb % 4 or ldb < -(-N // 4) * 4 or A.data_ptr() % 16 or B.data_ptr() % 16:
                raise capi.ErcGraftError("split wgrad: M=%d lda=%d N=%d ldb=%d unsupported" % (M, lda, N, ldb))
            self.deferred16.append((A, lda, B, ldb, Cm, ldc, M, N, K, bool(ct), bias_a, bias_b, gather, k_dev))
            return
        if A.dtype != torch.bfloat16 or B.dtype != torch.bfloat16 or Cm.dtype != torch.float32:
            raise capi.ErcGraftError("bf16 wgrad: operand dtypes %s %s %s" % (A.dtype, B.dtype, Cm.dtype))
        if M > 128 or lda % 8 or lda < -(-M // 8) * 8 or ldb % 4 or ldb < -(-N // 4) * 4 or A.data_ptr() % 16 or B.data_ptr() % 8:
            raise capi.ErcGraftError("bf16 wgrad: M=%d lda=%d N=%d ldb=%d unsupported" % (M, lda, N, ldb))
        self.deferred16.append((A, lda, B, ldb, Cm, ldc, M, N, K, bool(ct), bias_a, bias_b, gather, k_dev))

    def defer16_range(self, g_range):
        """fused optimizer only: ``g_range`` (a slice of the flat gradient) was completed by an earlier launch of the step;
        one work item of the weight-gradient launch applies the update to it"""
        self.ranges16.append(g_range)

    fused_adam = None     # a FusedAdam: its update is applied by the bf16 weight-gradient launch itself (single-rank steps)
    split_terms = 1       # 2 | 3: the records of defer16 hold fp32 operands (split compute modes f32x2 / f32x3)

    def flush_wgrads_bf16(self, cache):
        """Every record of defer16 as ONE launch (erc_wgrad_bf16); table, slabs and counters are built once per shape.
        With ``fused_adam`` set (and nothing it cannot do: clip-norm, a single identity shadow, a gradient exchange) the
        launch also applies the optimizer step -- ``adam_fused`` tells the trainer to skip ``optim.step()``."""
        if not self.deferred16:
            return
        opt = self.fused_adam
        # (after a timeout event of the fused launch -- FlatParams.check_health sets `tainted` -- the planner keeps the two-launch
        #  form, whose optimizer skips a step as a whole, for the rest of the run)
        # (data parallel: only with the peer-to-peer exchange, which then happens inside the launch -- erc_wgrad_adam_p2p)
        p2p = getattr(opt.flat, "p2p", None) if opt is not None else None
        fuse = opt is not None and opt.clip_norm <= 0 and opt.shadow is None and (p2p is not None or _world_size() == 1) and \
            opt.flat.grad is self.grad and not getattr(opt.flat, "tainted", False)
        import ctypes
        import struct
        key = tuple((a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, ct, g.data_ptr() if g is not None else 0,
                     kd.data_ptr() if kd is not None else 0) for a, _, b, _, c, _, M, N, K, ct, _, _, g, kd in self.deferred16) + \
            ((tuple((r.data_ptr(), r.numel()) for r in self.ranges16), ) if fuse else ())
        # large K (B = 512: N = 33 k): the wide form -- a workgroup's four wavefronts take four neighbouring column tiles over the
        # same k-steps, the A operand reaches a CU once per four tiles (csrc/wgrad_bf16.hip WIDE)
        terms = self.split_terms
        wide = terms == 1 and max(d[8] for d in self.deferred16) > int(os.environ.get("ERC_W2_WIDE_K", 8192))
        fuse = fuse and not wide
        if cache.get("w16_key") != key:
            cap = capi.wgrad_bf16_max_k_per_split()
            tiles = sum(-(-d[7] // 64) for d in self.deferred16)
            if wide:
                tiles = sum(-(-(-(-d[7] // 64)) // 4) for d in self.deferred16)      # workgroups per split
            K = max(d[8] for d in self.deferred16)
            # one 4-wavefront workgroup per CU (428 registers per lane): while every item of the launch is resident at once
            # (<= 256), as many splits as that allows; beyond, ~ERC_W2_ROWS k per item
            rows = int(os.environ.get("ERC_W2_ROWS", 1024))
            n_cu_all = min(256, torch.cuda.get_device_properties(self.device).multi_processor_count) if torch.cuda.is_available() else 256
            s_max = max(1, min(32, n_cu_all // tiles, -(-K // 64)))
            # a wavefront works in groups of 8 k-steps (32 k): among the split counts that fit, the smallest one with the
            # fewest groups per wavefront (K = 1982: 4 splits of 31 steps per wavefront, not 5 of 25 -- both are 4 groups)
            groups = lambda sp: -(-(-(-(-(-K // 4) // sp) // 4)) // 8)
            # (with the optimizer fused in, a tile's splits share its 8 quads per thread: powers of two only -- a K loop one
            #  group longer costs ~1 us, the optimizer launch it saves ~7)
            cand = [sp for sp in (1, 2, 4, 8) if sp <= s_max] if fuse else range(1, s_max + 1)
            splits = min(cand, key=lambda sp: (groups(sp), sp))
            splits = max(splits, min(32, -(-K // rows)), -(-K // cap))
            if wide:
                splits = max(-(-K // cap), max(1, min(32, 256 // tiles)))
            raw, items, n_tiles, bases, sps, whole_quads, wgs = [], 0, 0, [], set(), True, 0
            for a, lda, b, ldb, c, ldc, M, N, Kr, ct, ba, bb, g, kd in self.deferred16:
                nks = -(-Kr // 4)
                per = -(-nks // splits)
                sp = -(-nks // per)                       # no empty split
                if per * 4 > cap:
                    raise capi.ErcGraftError("bf16 wgrad: K=%d needs more than %d splits" % (Kr, splits))
                tn = -(-N // 64)
                cvec = int(c.data_ptr() % 16 == 0 and ldc % 4 == 0)
                n_wg = -(-tn // 4) * sp if wide else tn * sp      # workgroups of the record (its slabs: tn * sp either way)
                raw.append(struct.pack("<QQQQQQQ14i", a.data_ptr(), b.data_ptr(), c.data_ptr(),
                                       ba.data_ptr() if ba is not None else 0, bb.data_ptr() if bb is not None else 0,
                                       g.data_ptr() if g is not None else 0, kd.data_ptr() if kd is not None else 0,
                                       lda, ldb, ldc, M, N, Kr, int(ct), cvec, sp, tn, items, n_wg, n_tiles, 0))
                bases.append(wgs if wide else items)
                wgs += n_wg
                sps.add(sp)
                whole_quads = whole_quads and bool(cvec) and (M if ct else N) % 4 == 0
                items += tn * sp
                n_tiles += tn
            # the fused optimizer needs every work item resident at once (its splits wait for each other), split counts
            # that divide a thread's 8 quads, and gradients made of whole aligned quads
            # (one 4-wavefront workgroup per CU: the device's CU count, not a constant, bounds what is resident; the waits also
            #  rely on workgroups being dispatched in order, and are bounded)
            n_cu = min(256, torch.cuda.get_device_properties(self.device).multi_processor_count) if torch.cuda.is_available() else 256
            fused = fuse and items + len(self.ranges16) <= n_cu and sps <= {1, 2, 4, 8} and whole_quads
            if fused:     # finished gradient ranges (kind 1): one work item each
                for r in self.ranges16:
                    raw.append(struct.pack("<QQQQQQQ14i", 0, 0, r.data_ptr(), 0, 0, 0, 0, 0, 0, 0, r.numel(), 0, 0, 0, 0, 1, 1,
                                           items, 1, n_tiles, 1))
                    bases.append(items)
                    items += 1
            cache["w16_fused"] = fused
            cache["w16_tiles"] = n_tiles
            cache["w16_records"] = len(raw)
            # (the table is a small host -> device copy made OUTSIDE any stream capture: trainer.StepGraphs runs the first
            #  step of a shape on the graph's own static buffers before it captures)
            cache["w16_table"] = torch.frombuffer(bytearray(b"".join(raw)), dtype=torch.uint8).to(self.device)
            cache["w16_slabs"] = torch.empty(items * capi.wgrad_bf16_slab_floats(), dtype=torch.float32, device=self.device)
            cache["w16_counters"] = torch.zeros(n_tiles + 512, dtype=torch.int32, device=self.device)
            cache["w16_items"] = wgs if wide else items
            cache["w16_bases"] = (ctypes.c_int32 * len(bases))(*bases)
            cache["w16_key"] = key
        if fuse and cache["w16_fused"]:
            f = opt.flat
            capi.wgrad_bf16_adam(cache["w16_table"], cache["w16_records"], cache["w16_bases"], cache["w16_items"], cache["w16_slabs"],
                                 cache["w16_counters"], cache["w16_tiles"], f.data, f.grad, f.exp_avg, f.exp_avg_sq, f.numel, opt.lr, opt.betas[0],
                                 opt.betas[1], opt.eps, opt.weight_decay, opt.decoupled, 1.0 / p2p.world if p2p is not None else 1.0,
                                 opt.state, opt.shadow_table, opt.skip_flag, terms=terms, p2p_desc=p2p.desc if p2p is not None else None)
            self.adam_fused = True
            return
        if wide:
            capi.wgrad_bf16_wide(cache["w16_table"], cache["w16_records"], cache["w16_bases"], cache["w16_items"], cache["w16_slabs"],
                                 cache["w16_counters"])
        else:
            capi.wgrad_bf16(cache["w16_table"], cache["w16_records"], cache["w16_bases"], cache["w16_items"], cache["w16_slabs"],
                            cache["w16_counters"], terms=terms)

    def split_for(self, M, N, K, bk=None, min_chunks=None):
        if N <= 1025 and bk is None:
            return 1   # skinny output: the register-streaming kernel splits K inside the workgroup, no slabs
        bk = bk or self.BK
        min_chunks = self.MIN_CHUNKS if min_chunks is None else min_chunks
        tiles = -(-N // 32) * -(-M // 64)
        nchunk = -(-K // bk)
        want = max(1, -(-self.TARGET_WG // tiles))
        return max(1, min(want, self.MAX_SPLIT, nchunk // min_chunks if nchunk >= min_chunks else 1))

    def take(self, n):
        start = (self.cursor + ALIGN - 1) // ALIGN * ALIGN
        if start + n > self.ws.numel():
            raise capi.ErcGraftError("slab workspace too small (%d + %d > %d)" % (start, n, self.ws.numel()))
        self.cursor = start + n
        return start

    def add_job(self, src, stride, S, numel, dst_off):
        self.jobs.append((src, stride, S, numel, dst_off))
        self.max_numel = max(self.max_numel, numel)

    def job_table(self):
        return torch.tensor(self.jobs, dtype=torch.int64, device=self.device)

    def reduce_into(self, cache, grad):
        """Flush the deferred weight gradients, then run the batched slab reduce for the registered jobs (no-op
        when every gradient was written directly)."""
        self.flush_wgrads(cache)
        self.flush_wgrads_bf16(cache)
        if not self.jobs:
            return
        if cache.get("jobs") is None or cache["jobs"].shape[0] != len(self.jobs):
            cache["jobs"] = self.job_table()
        capi.slab_reduce_batched(self.ws, grad, cache["jobs"], len(self.jobs), self.max_numel)


def linear_fwd(pl, x, ldx, gather, W, bias, out, ldo, M, N, K, act=0, drop_p=0.0, rng=None, x_bf16=False, ldw=None):
    """out[M,N] (row pitch ldo) = act(x[M,K] @ W[N,K]^T + bias): nn.Linear forward.
    Split-K + slab reduce when K is long.  ``ldw`` = row pitch of W (column slice of a wider weight)."""
    ldw = K if ldw is None else ldw
    if x_bf16:
        assert act in (0, 1)
        capi.gemm_bf16a_stream(x, ldx, gather, W, ldw, out, ldo, M, N, K, bias=bias, act=act)
        return
    S = pl.split_for(M, N, K)
    if S == 1:
        capi.gemm_f32(x, ldx, 0, gather, W, ldw, 0, None, out, ldo, M, N, K, bias=bias, act=act,
                      act_scale=(1.0 / (1.0 - drop_p) if act == 3 else 1.0), drop_p=drop_p, rng_state=rng)
    else:
        assert act in (0, 1)
        src = pl.take(S * M * N)
        capi.gemm_f32(x, ldx, 0, gather, W, ldw, 0, None, pl.ws[src:], N, M, N, K, split_k=S, c_slab=M * N)
        capi.slab_reduce(pl.ws[src:], S, M * N, bias, N, act, out, M * N, ld_out=0 if ldo == N else ldo)


def linear_wgrad(pl, dy, lddy, x, ldx, gather, n_out, n_in, n_rows, w_off, b_off, x_bf16=False,
                 slab=None, col_off=0, ld_w=None, force_slab=False, defer=False):
    """dW[n_out,n_in] = dy^T x and db[n_out] = colsum(dy)  (nn.Linear weight layout [out,in]).
    Unsplit products are written straight into the flat gradient buffer; split ones go to partial slabs and
    register a reduce job.  ``slab`` (the return value of a previous call) lets several calls fill column
    slices [col_off, col_off+n_in) of ONE wider gradient (concatenated inputs).  ``defer``: postpone the product to
    the planner's batched launch (only legal when dy and x are not overwritten before GemmPlanner.flush_wgrads)."""
    ld_w = n_in if ld_w is None else ld_w
    want_b = b_off is not None
    if slab is None:
        S = pl.split_for(n_out, n_in + 1, n_rows, bk=64 if x_bf16 else None, min_chunks=2)
        direct = pl.grad is not None and not force_slab and (defer or (S == 1 and not x_bf16))
        slab = ("direct", w_off, ld_w) if direct else (pl.take(S * n_out * ld_w), S, ld_w)
        if not direct and w_off is not None:
            pl.add_job(slab[0], n_out * ld_w, S, n_out * ld_w, w_off)
    if slab[0] == "direct":
        _, base, ld_w = slab
        if defer:
            # a bf16 x operand (the feature block) takes the bf16 matrix-core path: dy is rounded too, as the split-K
            # bf16 GEMM this product used before did (the bias strip stays an fp32 sum)
            pl.defer(dy, lddy, x, ldx, pl.grad[base + col_off:], ld_w, n_out, n_in, n_rows, 1 if want_b else 0,
                     pl.grad[b_off:] if want_b else None, gather=gather, mma_bf16=True if x_bf16 else None)
        elif x_bf16:
            raise capi.ErcGraftError("bf16 column slice into a directly written gradient: pass force_slab=True")
        else:
            capi.gemm_f32(dy, lddy, 1, None, x, ldx, 1, gather, pl.grad[base + col_off:], ld_w, n_out, n_in, n_rows,
                          ones_col=1 if want_b else 0, bias_out=pl.grad[b_off:] if want_b else None)
        return slab
    src_w, S, ld_w = slab
    src_b = pl.take(S * n_out) if want_b else 0
    cbase = pl.ws[src_w + col_off:]
    if x_bf16:
        capi.gemm_bf16x(dy, lddy, 1, None, x, ldx, 1, gather, 0, cbase, ld_w, n_out, n_in, n_rows,
                        split_k=S, c_slab=n_out * ld_w, ones_col=1 if want_b else 0,
                        bias_out=pl.ws[src_b:] if want_b else None, bias_slab=n_out)
    else:
        capi.gemm_f32(dy, lddy, 1, None, x, ldx, 1, gather, cbase, ld_w, n_out, n_in, n_rows,
                      split_k=S, c_slab=n_out * ld_w, ones_col=1 if want_b else 0,
                      bias_out=pl.ws[src_b:] if want_b else None, bias_slab=n_out)
    if want_b:
        pl.add_job(src_b, n_out, S, n_out, b_off)
    return slab


def matmul_wgrad_io(pl, x, ldx, dy, lddy, n_in, n_out, n_rows, w_off, b_off, defer=False, scale=1.0):
    """dW[n_in,n_out] = x^T dy and db[n_out] = colsum(dy) for [in,out]-stored weights (PyG RGCNConv, GCNII)."""
    want_b = b_off is not None
    S = pl.split_for(n_in + 1, n_out, n_rows, min_chunks=2)
    if (S == 1 or defer) and pl.grad is not None:
        if defer:
            pl.defer(x, ldx, dy, lddy, pl.grad[w_off:], n_out, n_in, n_out, n_rows, 2 if want_b else 0,
                     pl.grad[b_off:] if want_b else None, scale=scale)
        else:
            capi.gemm_f32(x, ldx, 1, None, dy, lddy, 1, None, pl.grad[w_off:], n_out, n_in, n_out, n_rows,
                          ones_col=2 if want_b else 0, bias_out=pl.grad[b_off:] if want_b else None)
        return
    src_w = pl.take(S * n_in * n_out)
    src_b = pl.take(S * n_out) if want_b else 0
    capi.gemm_f32(x, ldx, 1, None, dy, lddy, 1, None, pl.ws[src_w:], n_out, n_in, n_out, n_rows,
                  split_k=S, c_slab=n_in * n_out, ones_col=2 if want_b else 0,
                  bias_out=pl.ws[src_b:] if want_b else None, bias_slab=n_out)
    pl.add_job(src_w, n_in * n_out, S, n_in * n_out, w_off)
    if want_b:
        pl.add_job(src_b, n_out, S, n_out, b_off)


class FusedAdam:
    """torch.optim.Adam / AdamW semantics on a FlatParams buffer, one launch."""

    def __init__(self, flat, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False,
                 clip_norm=0.0, seed=1):
        self.flat, self.lr, self.betas, self.eps = flat, lr, betas, eps
        self.weight_decay, self.decoupled, self.clip_norm = weight_decay, decoupled, clip_norm
        dev = flat.device
        # {step, rng offset, rng seed}
        # {step, rng offset, rng seed, unused, one private copy of the step count per workgroup of the optimizer launch}
        self.state = torch.zeros(4 + 512, dtype=torch.int64, device=dev)
        self.state[2] = seed
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.norm_ws = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.shadow = None   # (bf16 tensor, offset, numel): kept in sync with the fp32 master weights by the step
        self.shadow_table = None   # capi.ShadowTable: several bf16 ranges / layouts written by the same launch
        self.skip_flag = None  # device int32: non-zero = this step's gradients are invalid, the kernel skips the update

    def set_step(self, step):
        """step count by hand (checkpoint load): state[0] and the workgroups' private copies"""
        self.state[0] = int(step)
        self.state[4:] = int(step)

    @property
    def rng_state(self):
        return self.state[1:3]

    def enable_p2p(self, group=None):
        """ERC_DP_P2P=1 and a process group of >= 2 ranks: fuse the gradient exchange into this optimizer's launch
        (P2PExchange).  Collective call; returns whether it is on.  Modules with clip-norm or more than 524 288
        parameters keep the RCCL all-reduce."""
        import torch.distributed as dist
        if os.environ.get("ERC_DP_P2P", "0") != "1" or not (dist.is_available() and dist.is_initialized()) or \
                dist.get_world_size(group) < 2 or self.clip_norm > 0 or self.flat.numel > P2PExchange.MAX_PARAMS:
            return False
        try:
            self.flat.p2p = P2PExchange(self.flat, group)
        except P2PUnavailable as exc:      # (raised on every rank alike: the decision rides an all_gather)
            import sys
            print("ERC_DP_P2P=1: %s -- keeping the RCCL all-reduce" % exc, file=sys.stderr)
            return False
        self.skip_flag = self.flat.health
        return True

    def step(self, grad_scale=1.0):
        f = self.flat
        if getattr(f, "p2p", None) is not None:
            capi.adam_step_p2p(f.data, f.grad, f.exp_avg, f.exp_avg_sq, f.numel, self.lr, self.betas[0], self.betas[1], self.eps,
                               self.weight_decay, self.decoupled, grad_scale, self.state, self.shadow_table, f.p2p.desc)
            return
        if self.clip_norm > 0:
            capi.grad_norm(f.grad, f.numel, grad_scale, self.gnorm, self.norm_ws)
        if self.shadow_table is not None:
            capi.adam_step_tab(f.data, f.grad, f.exp_avg, f.exp_avg_sq, f.numel, self.lr, self.betas[0], self.betas[1],
                               self.eps, self.weight_decay, self.decoupled, grad_scale, self.clip_norm,
                               self.gnorm if self.clip_norm > 0 else None, self.state, self.shadow_table,
                               skip_flag=self.skip_flag)
            return
        capi.adam_step(f.data, f.grad, f.exp_avg, f.exp_avg_sq, f.numel, self.lr, self.betas[0], self.betas[1],
                       self.eps, self.weight_decay, self.decoupled, grad_scale, self.clip_norm,
                       self.gnorm if self.clip_norm > 0 else None, self.state,
                       *(self.shadow if self.shadow is not None else (None, 0, 0)), skip_flag=self.skip_flag)


class P2PUnavailable(RuntimeError):
    """the fused gradient exchange cannot be set up on this node (every rank raises it together)"""


class P2PExchange:
    """ERC_DP_P2P=1 (OFF by default; UNVERIFIED ACROSS DEVICES: it has only ever run as two processes on one GPU,
    tests/test_gpu_p2p.py -- the RCCL all-reduce stays the default until a multi-GPU run exists): the gradient exchange of a data-parallel step fused into the optimizer launch (csrc/optim.hip, P2PArgs;
    SURVEY.md 8e).  Construction is collective: every rank allocates its publish buffer and flag array, the 64-byte IPC
    handles travel through ``torch.distributed.all_gather_object`` (any backend), peers are mapped.  ``FusedAdam.step``
    then calls erc_adam_step_p2p instead of all-reduce + erc_adam_step: no RCCL call, no extra launch."""

    MAX_PARAMS = 512 * 1024

    def __init__(self, flat, group=None):
        import ctypes as C
        import torch.distributed as dist
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if self.world > 8 or flat.numel > self.MAX_PARAMS or flat.numel % 4:
            raise capi.ErcGraftError("P2P gradient exchange: world <= 8 and <= %d parameters (this module: %d)" % (
                self.MAX_PARAMS, flat.numel))
        self.n_pad = flat.numel
        self._mine, self._peers = [], []
        try:      # uncached (fine-grained) device memory or nothing: capi.p2p_alloc has no coarse-grained fallback
            self._mine = [capi.p2p_alloc(2 * self.n_pad * 4), capi.p2p_alloc(self.world * 512 * 4)]
            mine = (self._mine[0][1], self._mine[1][1])
        except capi.ErcGraftError as exc:
            mine = None
            self.alloc_error = str(exc)
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        if any(h is None for h in handles):      # collective decision: one rank without the memory -> every rank keeps RCCL
            self.close()
            raise P2PUnavailable("rank(s) %s could not allocate uncached peer-visible memory" % [r for r, h in enumerate(handles) if h is None])
        self.allocation = "hipDeviceMallocUncached"
        x = capi.ErcP2P()
        x.world, x.rank, x.spin_limit, x.n_pad = self.world, self.rank, int(os.environ.get("ERC_P2P_SPIN", 0)), self.n_pad
        for r in range(self.world):
            if r == self.rank:
                x.pub[r], x.flags[r] = self._mine[0][0], self._mine[1][0]
            else:
                pp, pf = capi.p2p_open(handles[r][0]), capi.p2p_open(handles[r][1])
                self._peers += [pp, pf]
                x.pub[r], x.flags[r] = pp, pf
        self.epoch = torch.zeros(512, dtype=torch.int64, device=flat.device)
        x.epoch, x.health = self.epoch.data_ptr(), flat.health.data_ptr()
        self.desc, self.flat = x, flat
        dist.barrier(group=group)          # nobody launches before everybody has mapped everybody

    def close(self):
        for p in self._peers:
            capi.p2p_close(p)
        self._peers = []
        for p, _ in self._mine:
            capi.p2p_free(p)
        self._mine = []


def all_reduce_grads(flat, always=False):
    """DP exchange step: one sum all-reduce of the flat live-gradient buffer;
    the 1/world scaling is folded into the optimizer's grad_scale.  `always` issues the collective on a
    one-rank group too (bench.py --rehearse_dp)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or always):
        if getattr(flat, "p2p", None) is not None:
            return 1.0 / dist.get_world_size()      # ERC_DP_P2P=1: the optimizer launch sums the ranks' gradients itself
        dist.all_reduce(flat.grad_full)     # gradients + the health word behind them
        return 1.0 / dist.get_world_size()
    return 1.0


class SideStream:
    """Fork / join helper: weight-gradient GEMMs are off the critical path of the backward chain (nothing but the
    optimizer consumes them), so they are enqueued on a second stream and overlap the next stages.  Works the same
    eagerly and under HIP-graph capture (event fork/join becomes graph dependencies)."""

    def __init__(self, enabled=None):
        # measured on MI355X / ROCm 7.2: replayed HIP graphs run the forked branch no earlier than the main one
        # (step 299 us forked vs 282 us in-line), so the fork is opt-in (ERC_SIDE_STREAM=1)
        if enabled is None:
            import os
            enabled = os.environ.get("ERC_SIDE_STREAM", "0") == "1"
        self.enabled = enabled and torch.cuda.is_available()
        self.stream = torch.cuda.Stream() if self.enabled else None
        self._dirty = False

    def fork(self):
        return _Fork(self)

    def join(self):
        if self.enabled and self._dirty:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            torch.cuda.current_stream().wait_event(ev)
            self._dirty = False


class _Fork:
    def __init__(self, side):
        self.side = side

    def __enter__(self):
        if self.side.enabled:
            ev = torch.cuda.Event()
            ev.record()
            self.side.stream.wait_event(ev)
            self.ctx = torch.cuda.stream(self.side.stream)
            self.ctx.__enter__()
            self.side._dirty = True

    def __exit__(self, *exc):
        if self.side.enabled:
            self.ctx.__exit__(*exc)
        return False


class GraphedStep:
    """Capture ``fn()`` (a whole training step whose every launch goes to the current stream and which
    performs no host synchronisation) into one HIP graph and replay it.  The inputs of ``fn`` must live in
    fixed device buffers; shapes are static per captured graph (one graph per (B, T, N) bucket)."""

    def __init__(self, fn, warmup=2, steps=1):
        """``steps`` > 1 captures that many consecutive calls of ``fn`` into the one graph (a loop whose next batches are already
        resident: the ~5.5 us bubble between two graph launches is paid once per ``steps`` steps)."""
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):      # allocates workspaces / job tables outside the capture
                self.out = fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            for _ in range(steps):
                self.out = fn()
        self.warmup_steps, self.steps = warmup, steps

    def __call__(self):
        self.graph.replay()
        return self.out
