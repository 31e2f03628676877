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
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFS = 157.3   # v_mfma_f32_16x16x4_f32 / 32x32x2_f32: exact fp32, = the fp32 vector rate (same guide)
MFMA_BF16_PEAK_TFS = 2500.0  # dense bf16
PROFILE_ROUND = "r04"        # profiles/<round>_<module>_b<B>_<dtype>_pmc.json: counter summaries of this same command


def step_work(module, params, host_batch, features_bf16):
    """Algorithmic work of ONE training step (fwd + bwd + update, live path only): SURVEY.md 8(d) per-utterance table x
    the units of this batch.  Returns (bytes, flops, note).  COGMEN / MMGCN / DialogueGCN count the N valid utterances;
    DAG-ERC counts all B*T padded positions for the FLOPs (the reference runs the recurrence over the padded length,
    dagerc.py:167-189) and the valid ones for the bytes."""
    N = int(host_batch["label"].shape[0])
    D = params.hidden_all
    if module == "cogmen":
        row = D * (2 if features_bf16 else 4)
        per_utt = 2 * row + 4800 + 180                 # feature row read in fwd and in wgrad; 12 x 100-wide fp32; edges
        live = 100 * D + 100 + 90100 + 40400 + 200 + 10100 + 101 * params.n_classes
        return per_utt * N + live * 16, 1.4e6 * N * (D / 1380.0 * 0.59 + 0.41), "N=%d utterances; 10.5 KB + 1.4 MFLOP per utterance at D=1380, %.2f MB of parameters + Adam state" % (N, live * 16 / 1e6)
    if module == "dagerc":
        B, T = host_batch["input_tensor"].shape[:2]
        live = 6427510 if D == 1380 else 6026710 if D == 712 else (300 * D + 300 + 4 * (4 * 541800 + 180000 + 601) + 300 * (1500 + D) + 300 + 90300 + 301 * params.n_classes)
        flops_pos = 3.0 * (2 * D * 300 + 4 * 2.55e6 + 2 * (1500 + D) * 300 + 2 * 300 * 300)
        return (2 * D * (2 if features_bf16 else 4) + 48000) * N + live * 16, flops_pos * B * T, \
            "%d padded positions x %.1f MFLOP, %d valid utterances x 53.6 KB, %.1f MB of parameters + Adam state" % (
                B * T, flops_pos / 1e6, N, live * 16 / 1e6)
    if module == "mmgcn":
        live = 5927606
        return (1380 * 4 + 460000) * N + live * 16, 111e6 * N, \
            "N=%d utterances x (477 KB with the 64 per-layer activations stored for the backward, 111 MFLOP)" % N
    live = 2030000    # dgcn, MELD atv
    lstm_act = (2 * 800 + 2 * 200 + 2 * 200) * 4 * 2
    fl = 3.0 * (2 * 2 * 4 * 100 * (D + 100) + 2 * 2 * 4 * 100 * 300 + 88e3 + 0.84e6 + 62e3)
    return (2 * D * (2 if features_bf16 else 4) + lstm_act + 420) * N + live * 16, fl * N, \
        "N=%d utterances x (%.1f KB, %.1f MFLOP)" % (N, (2 * D * 4 + lstm_act + 420) / 1e3, fl / 1e6)


# entry points whose kernels can dominate a step, per module, with their algorithmic FLOPs per launch
def _probe_candidates(module, params, host_batch, trainer):
    lens = host_batch["text_length"].tolist()
    N = sum(lens)
    if module == "cogmen":
        pl = trainer.model._last_ws["planner"]
        recs = pl.flushed + pl.deferred + pl.deferred16
        fl = sum(2.0 * d[6] * d[7] * d[8] for d in recs)
        # operands of every dW = A^T B record read once + the gradient written once: K*M*sizeof(A) + K*N*sizeof(B) + M*N*4
        wg_bytes = sum(d[8] * d[6] * d[0].element_size() + d[8] * d[7] * d[2].element_size() + d[6] * d[7] * 4.0
                       for d in recs)
        C = params.n_classes
        E = 11 * N
        fused = bool(trainer.model._last_ws.get("fused"))
        bf = (MFMA_BF16_PEAK_TFS, "bf16 matrix-core peak (v_mfma_f32_16x16x32_bf16, dense): the products of this kernel run on bf16 "
                                  "operands with fp32 accumulation (bf16 compute mode)")
        f32p = (MFMA_F32_PEAK_TFS, None)
        # algorithmic bytes per launch = what the launch must read and write once
        head_b = N * 400.0 * 6 + 2 * 40400 + 101 * C * 4       # H2 in; H3, Z, dZ, dH3 + logits / dlogits out; cls weights
        fwd_b = N * (400 + 1600 + 1808 + 208 + 400) + E * (8 + 4.0) + 0.26e6   # H0 in; QKVS, bf16 M / H1, H2, alpha out; CSR; shadows
        bwd_b = N * (400 + 400 + 1600 + 1600 + 400 + 400) + E * (17 + 4.0) + 0.26e6   # dY, H2, QKVS in; dQKVS, dH1, dH0 out
        cands = {"erc_wgrad_table": ("wgrad_table_kernel (every weight gradient of the step, one launch)", fl, wg_bytes) +
                 (bf if fused and trainer.model.wgrad_bf16 else f32p),
                 "erc_wgrad_bf16": ("wgrad_bf16_kernel (every weight gradient of the step from bf16 operands, one launch)", fl,
                                    wg_bytes) + bf,
                 # the same launch with the optimizer fused in: + parameters and both moments read and written once, the bf16
                 # shadows written once
                 "erc_wgrad_bf16_adam": ("wgrad_bf16_kernel<adam> (every weight gradient of the step from bf16 operands + the "
                                         "Adam update of every parameter, one launch)", fl,
                                         wg_bytes + trainer.model.flat.numel * 24.0 +
                                         (trainer.model.shadows.buf.numel() * 2.0 if trainer.model.shadows is not None else 0.0)) + bf,
                 # BatchNorm apply, two 100 x 100 products forward + one backward, the C-wide products on the VALU
                 "erc_head_fused": ("head_fused_kernel (BatchNorm apply .. cross entropy .. dY, one launch)",
                                    N * (3 * 2.0 * 100 * 100 + 4.0 * 100 * C), head_b) + f32p,
                 "erc_head_fused_bn": ("head_fused_kernel<bn> (BatchNorm statistics + apply .. cross entropy .. dY, one launch)",
                                       N * (3 * 2.0 * 100 * 100 + 4.0 * 100 * C), head_b) + f32p,
                 # algorithmic products of the graph part (no halo recomputation counted): M Wcat, H1 Wq | dQKVS Wq, dP Wb
                 "erc_cogmen_fwd_tile": ("cogmen_fwd_tile_kernel (relation means, RGCN and QKVS products, attention; halo tiles)",
                                         N * 2.0 * (900 * 100 + 100 * 400), fwd_b) + bf,
                 "erc_cogmen_bwd_tile": ("cogmen_bwd_tile_kernel (BatchNorm / attention backward, dH1 and dH0 products; halo tiles)",
                                         N * 2.0 * (400 * 100 + 900 * 100), bwd_b) + bf}
        nt = getattr(trainer.model, "terms", 1)
        if nt > 1:
            ntb = trainer.model.terms_bwd
            # split compute modes: the same launches on fp32 data; a product costs nt (nt + 1) / 2 term products on the bf16 matrix
            # cores, counted here as ALGORITHMIC fp32 FLOPs against the rate those term products allow (bf16 peak / their number)
            def sp_of(n_):
                npr = n_ * (n_ + 1) // 2
                return (MFMA_BF16_PEAK_TFS / npr, "bf16 matrix-core peak / %d: every fp32-class product is %d term products on "
                                                   "v_mfma_f32_16x16x32_bf16 (%d bf16 terms per operand value)" % (npr, npr, n_))
            sp, spb = sp_of(nt), sp_of(ntb)
            fwd_x = N * (400 + 1600 + 3600 + 400 + 400) + E * (8 + 4.0) + 0.26e6 * nt      # H0 in; QKVS, fp32 M / H1, H2, alpha out
            bwd_x = N * (400 + 400 + 1600 + 1600 + 400 + 400) + E * (17 + 4.0) + 0.26e6 * nt
            adam_b = trainer.model.flat.numel * 24.0 + (trainer.model.shadows.buf.numel() * 2.0 if trainer.model.shadows is not None else 0.0)
            cands.update({
                "erc_wgrad_split": ("wgrad_bf16_kernel<terms=%d> (every weight gradient of the step from fp32 operands, one launch)" % ntb,
                                    fl, wg_bytes) + spb,
                "erc_wgrad_split_adam": ("wgrad_bf16_kernel<adam, terms=%d> (every weight gradient of the step from fp32 operands + the "
                                         "Adam update of every parameter, one launch)" % ntb, fl, wg_bytes + adam_b) + spb,
                "erc_cogmen_fwd_tile_x": ("cogmen_fwd_tile_kernel<terms=%d> (relation means, RGCN and QKVS products, attention; halo tiles)" % nt,
                                          N * 2.0 * (900 * 100 + 100 * 400), fwd_x) + sp,
                "erc_cogmen_bwd_tile_x": ("cogmen_bwd_tile_kernel<terms=%d> (BatchNorm / attention backward, dH1 and dH0 products; halo tiles)" % ntb,
                                          N * 2.0 * (400 * 100 + 900 * 100), bwd_x) + spb})
        return cands
    if module == "dagerc":
        B, T = host_batch["input_tensor"].shape[:2]
        per = 2.0 * 300 * (1800 + 600 + 1)
        L = params.get("gnn_layers", 4)
        # per position and layer: hoisted [1801 x 300] + sequential [1800 x 300] + relations [601 x 300] products, forward;
        # the backward runs the three transposed products (the weight-gradient GEMMs are separate launches)
        per = 2.0 * 300 * (1801 + 1800 + 601)
        # weights once per launch + per position and layer: input row, hoisted gates (1801), sequential gates (1800), Mseq,
        # relation sums (600), output row (forward writes / backward reads them, + the gradient rows it writes)
        wts = L * 300.0 * (1801 + 1800 + 601) * 4
        pos_f = (300 + 1801 + 1800 + 300 + 600 + 300) * 4.0
        pos_b = pos_f + (1801 + 1800 + 300 + 300) * 4.0
        f32p = (MFMA_F32_PEAK_TFS, None)
        return {"erc_dag_rec_fwd": ("dag_rec_fwd_kernel (weight-stationary forward recurrence, all %d layers pipelined)" % L,
                                    per * B * T * L, wts + pos_f * B * T * L) + f32p,
                "erc_dag_rec_bwd": ("dag_rec_bwd_kernel (weight-stationary backward recurrence, all %d layers pipelined)" % L,
                                    per * B * T * L, wts + pos_b * B * T * L) + f32p}
    if module == "mmgcn":
        Mo = len(params.modality)
        fl = sum(Mo * 2.0 * L * L * 200 for L in lens)
        f32p = (MFMA_F32_PEAK_TFS, None)
        adj = sum(Mo * 4.0 * L * L for L in lens)
        # per layer: the layer's V (200 x 200), c_l rows in, h / z rows out (forward) | h in, dg / dz out (backward)
        chain_f = adj + 64 * (160e3 + Mo * N * 800.0 * 3)
        chain_b = adj + 64 * (160e3 + Mo * N * 800.0 * 3)
        return {"erc_gemm_f32_grouped": ("gemm_f32_grouped_kernel<0> (A.h of one GCNII layer, per-dialogue blocks)", fl,
                                         adj + Mo * N * 1600.0) + f32p,
                "erc_gcnii_chain_fwd": ("gcnii_chain_fwd_kernel (64 GCNII layers, one launch)", 64 * (fl + 3 * N * 2.0 * 400 * 200), chain_f) + f32p,
                "erc_gcnii_chain_bwd": ("gcnii_chain_bwd_kernel (64 GCNII layers backward, one launch)", 64 * (fl + 3 * N * 2.0 * 2 * 400 * 200), chain_b) + f32p}
    per = 2 * 2.0 * 100 * 400
    f32p = (MFMA_F32_PEAK_TFS, None)
    # per position and direction pair: 800 hoisted gate pre-activations in, 200 outputs + 800 gates + 200 cell states saved
    return {"erc_lstm_scan_fwd": ("lstm_fwd_kernel (BiLSTM recurrence, one layer)", per * N, N * 4.0 * (800 + 200 + 800 + 200) + 320e3) + f32p,
            "erc_lstm_scan_bwd": ("lstm_bwd_kernel (BiLSTM recurrence backward, one layer)", per * N, N * 4.0 * (800 + 200 + 200 + 800) + 320e3) + f32p}


def time_replays(entries, reps):
    """Average duration of `reps` back-to-back launches (captured once, replayed as one HIP graph -- eager ctypes calls
    would time the interpreter) between one HIP event pair on the launching stream."""
    from erc_amd import capi
    for e in entries:
        capi.replay(e)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            for e in entries:
                capi.replay(e)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def dominant_kernel(module, params, host_batch, batch, trainer, reps):
    """Record one eager step, replay each candidate entry point on its own with the step's operands, and name the one
    with the largest share of the step (average launch duration x launches per step)."""
    from erc_amd import capi
    capi.start_recording()
    trainer.train_step(batch)
    rec = capi.stop_recording()
    torch.cuda.synchronize()
    cands = _probe_candidates(module, params, host_batch, trainer)
    best = None
    for name, cand in cands.items():
        label, flops, nbytes = cand[:3]
        calls = [e for e in rec if e[0] == name]
        if not calls:
            continue
        us = time_replays([calls[0]], max(10, reps // (10 if module in ("dagerc", "mmgcn") else 1)))
        share = us * len(calls)
        if best is None or share > best["share_us"]:
            best = {"entry": name, "kernel": label, "avg_us": us, "launches_per_step": len(calls), "share_us": share,
                    "flops": flops, "bytes": nbytes, "peak": cand[3], "peak_note": cand[4]}
    return best


def synthetic_batch(params, B, T, seed):
    from erc_amd.collate import ERCCollate
    from erc_amd.synthetic import make_dialogues
    dialogs = make_dialogues(B, params.dims(), n_speakers=params.n_speakers, n_classes=params.n_classes,
                             min_len=20 if T >= 40 else 1, max_len=T, seed=seed, force_max=True)
    batch = ERCCollate(params)([[d] for d in dialogs])
    batch.pop("utterance_texts", None)
    return batch


def _time_cpu(step, n_utt, budget_s, max_steps=10):
    step()  # warm-up
    times, t_all = [], time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_all < budget_s and len(times) < max_steps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    return n_utt / times[len(times) // 2], len(times)


def cpu_baseline(module, params, batch, budget_s=20.0):
    """The reference CPU path = the oracle (structure-faithful PyTorch-CPU restatement: dead encoder, per-edge
    python graph construction, per-step torch.cat regrowth, dense (3N)^2 adjacency ...) timed on this box's host
    cores, on a bounded sample."""
    n_utt = int(batch["label"].shape[0])
    torch.manual_seed(1)
    if module == "cogmen":
        from oracle.cogmen import COGMENOracle, cogmen_train_step
        out = {}
        for tag, dead in (("with_dead_encoder", True), ("without_dead_encoder", False)):
            model = COGMENOracle(params.hidden_all, 100, 17, params.n_speakers, params.n_classes, dead_encoder=dead)
            model.train()
            opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-8)
            out[tag] = _time_cpu(lambda: cogmen_train_step(model, opt, batch), n_utt, budget_s / 2)
        return out
    if module == "dagerc":
        from oracle.dagerc import DAGERCOracle, dagerc_train_step
        model = DAGERCOracle(emb_dim=params.hidden_all, dropout=params.get("dropout", 0.0), n_classes=params.n_classes)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        step = lambda: dagerc_train_step(model, opt, batch)
    elif module == "dgcn":
        from oracle.dgcn import DGCNOracle, dgcn_train_step
        model = DGCNOracle(params.n_speakers, input_size=params.hidden_all, n_classes=params.n_classes)
        opt = torch.optim.Adam(model.parameters(), lr=3e-4)
        step = lambda: dgcn_train_step(model, opt, batch)
    else:
        from oracle.mmgcn import MMGCNOracle, mmgcn_train_step
        model = MMGCNOracle(hidden_text=params.hidden_text, hidden_visual=params.hidden_visual,
                            hidden_audio=params.hidden_audio, n_speakers=params.n_speakers, n_classes=params.n_classes,
                            modals=params.modality)
        opt = torch.optim.Adam(model.parameters(), lr=3e-4, weight_decay=3e-5)
        step = lambda: mmgcn_train_step(model, opt, batch)
    model.train()
    return {"with_dead_encoder": _time_cpu(step, n_utt, budget_s, max_steps=5)}


# module -> (default dataset, default per-GPU batch, max length, extra flags): BASELINE.json configs[1..4]
WORKLOADS = {
    "cogmen": ("iemocap-cogmen-sbert-6", 32, 110, []),
    "mmgcn": ("iemocap-cogmen-sbert-6", 16, 110, []),
    "dagerc": ("iemocap-cogmen-6", 16, 110, ["--reimplement"]),
    "dgcn": ("meld-mmgcn-7", 32, 33, ["--loss_weights=False"]),
}


def assert_no_skipped_steps(model, what):
    """A number measured over steps whose update was skipped (a bounded wait of a persistent kernel ran into its bound) or only
    partly applied is not a measurement of the step: no line, rc 4."""
    if not hasattr(model, "check_cluster"):
        return
    try:
        model.check_cluster()
    except Exception as exc:
        print("bench.py: %s -- %s: %s" % (what, type(exc).__name__, exc), file=sys.stderr)
        sys.stderr.flush()
        os._exit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f32x2", "f32x3", "f32x32"],
                    help="compute mode: bf16 (BASELINE.json configs[1]); f32x3 / f32x2: fp32 data, products on the bf16 matrix cores from "
                         "operands expanded into 3 / 2 bf16 terms (COGMEN: the 1e-4 parity path on the fused step); f32: exact-fp32 kernels")
    ap.add_argument("--module", default="cogmen", choices=sorted(WORKLOADS))  # headline = cogmen (configs[1])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--max_len", type=int, default=None)
    ap.add_argument("--dataset", default=None)  # cogmen default: d_t=768 -> D=1380 as BASELINE.json config 2
    ap.add_argument("--modality", default="atv", help="subset of a / t / v (BASELINE.json configs[4]: DialogueGCN ablation)")
    ap.add_argument("--no_graph", action="store_true")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_fp32_path", action="store_true",
                    help="COGMEN bf16: skip the second timed run of the same step in --dtype f32 (the 1e-4 parity path)")
    ap.add_argument("--kernel_reps", type=int, default=200)
    ap.add_argument("--clock_probe", action="store_true", help="diagnostic: append a clock-probe kernel to the step")
    ap.add_argument("--faithful_dead_encoder", action="store_true",
                    help="COGMEN: also run the reference's dead Transformer encoder every step (SURVEY.md 8a C2 (ii))")
    ap.add_argument("--chained_encoder", action="store_true",
                    help="COGMEN: opt-in variant rnn.1(rnn.0(x, padding mask)) -- the encoder is trained (SURVEY.md 8f-4)")
    ap.add_argument("--rehearse_dp", action="store_true",
                    help="diagnostic: run the N>1 step structure (RCCL group, eager exchange + optimizer) on one rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL; gloo: --dry_run)")
    ap.add_argument("--dry_run", action="store_true",
                    help="launch the ranks, form the process group, count them with one all-reduce and print the line -- no GPU work "
                         "(the launcher's CPU test: --gpus 2 --backend gloo --dry_run)")
    args = ap.parse_args()

    # ---------------------------------------------------------------- N > 1 without a launcher: start the N ranks ourselves
    # (the reference gets its ranks from `accelerate launch`, lumo/trainer/trainer.py:62-64,377-384).  The parent never touches
    # the GPU (torch.cuda.device_count() does not initialise it): it re-runs this file under torch.distributed.run -- one fresh
    # process per GPU -- forwards their output (rank 0 prints the JSON line) and exits with their return code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        if not (args.dry_run and args.backend == "gloo"):
            have = torch.cuda.device_count()
            if have < args.gpus:
                print("bench.py: --gpus %d but only %d device(s) visible: not running (a 1-rank line would not be the requested "
                      "measurement)" % (args.gpus, have), file=sys.stderr)
                sys.exit(2)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dp = world > 1 or args.rehearse_dp
    if dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29573")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.dry_run:
            dist.init_process_group(args.backend)
            cnt = torch.ones(1, dtype=torch.float64, device="cuda:%d" % local_rank if args.backend == "nccl" else "cpu")
            dist.all_reduce(cnt)
            if rank == 0:
                print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": int(cnt.item()), "backend": args.backend,
                                  "requested_gpus": args.gpus}))
            dist.destroy_process_group()
            sys.exit(0 if int(cnt.item()) == args.gpus else 3)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(args.backend, device_id=torch.device("cuda", local_rank))
    if args.gpus != world and not args.rehearse_dp:
        # (a launcher that started a different number of ranks than --gpus asks for: refuse rather than mislabel the line)
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    import importlib
    from erc_amd import capi
    from erc_amd.engine import GraphedStep, all_reduce_grads
    capi.lib()  # fail loudly if the HIP library is missing

    ds, bsz, mlen, extra = WORKLOADS[args.module]
    args.dataset, args.batch, args.max_len = args.dataset or ds, args.batch or bsz, args.max_len or mlen
    if args.module == "mmgcn" or (args.module != "cogmen" and args.dtype.startswith("f32x")):
        args.dtype = "f32"  # three separate feature blocks; no bf16 feature mode for MMGCN; the split modes are COGMEN's
    plugin = importlib.import_module("track_mm." + args.module)
    if args.faithful_dead_encoder and args.module == "cogmen":
        extra = extra + ["--faithful_dead_encoder"]
    if args.chained_encoder and args.module == "cogmen":
        extra = extra + ["--chained_encoder"]
    params = plugin.ParamsType().from_args(["--dataset=" + args.dataset, "--modality=" + args.modality, "--compute=" + args.dtype]
                                           + extra)
    params.train.batch_size = args.batch
    trainer = getattr(plugin, {"cogmen": "COGMENTrainer", "mmgcn": "MMGCNTrainer", "dagerc": "DAGERCTrainer",
                               "dgcn": "DGCNTrainer"}[args.module])(params, device)
    host_batch = synthetic_batch(params, args.batch, args.max_len, seed=1 + rank)
    batch = trainer.prepare_batch(host_batch)
    n_utt = int(host_batch["label"].shape[0])

    # ---------------------------------------------------------------- step function
    use_graph = not args.no_graph
    probe = torch.zeros(4, dtype=torch.int64, device=device) if args.clock_probe else None
    if not dp:
        def step_fn():
            out = trainer.train_step(batch)
            if probe is not None:
                capi.clock_probe(probe, 400)
            return out
        step = GraphedStep(step_fn) if use_graph else step_fn
    else:
        # N > 1: the WHOLE step -- forward, backward, the RCCL sum all-reduce of the flat gradient buffer and the
        # optimizer -- is one captured HIP graph (RCCL collectives are capturable: the all-reduce becomes a graph node
        # behind the last weight-gradient kernel); ERC_DP_EAGER=1 keeps the exchange and the optimizer outside the graph
        # (the round-1 structure) for comparison
        trainer.model.train()
        if hasattr(trainer.model, "fused_optim") and getattr(trainer.model.flat, "p2p", None) is None:
            trainer.model.fused_optim = None      # N > 1 structure: the optimizer is its own launch behind the RCCL exchange
        # (ERC_DP_P2P=1: the exchange happens inside the weight-gradient + optimizer launch -- the step stays 5 launches)
        cw = getattr(trainer, "class_weight", None)
        dead_encoder = getattr(trainer, "encoder", None)            # --faithful_dead_encoder
        trained_encoder = getattr(trainer.model, "enc_train", None)  # --chained_encoder
        dp_eager = os.environ.get("ERC_DP_EAGER", "0") == "1"

        def fb():
            if dead_encoder is not None:
                dead_encoder.forward(batch["input_tensor"])
            if args.module in ("cogmen", "dgcn"):
                return trainer.model.loss_and_grads(batch, cw)
            return trainer.model.loss_and_grads(batch)

        def exchange_and_update():
            pl_ = trainer.model._last_ws.get("planner") if isinstance(getattr(trainer.model, "_last_ws", None), dict) else None
            if pl_ is not None and getattr(pl_, "adam_fused", False):
                return                                # the weight-gradient launch exchanged the gradients and applied the update
            trainer.optim.step(grad_scale=all_reduce_grads(trainer.model.flat, always=args.rehearse_dp))
            if trained_encoder is not None:
                trained_encoder.refresh_shadows()      # bf16 copies of the encoder weights follow the fp32 masters

        if dp_eager or not use_graph:
            fb_g = GraphedStep(fb) if use_graph else fb

            def step():
                out = fb_g()
                exchange_and_update()
                return out
        else:
            def whole():
                out = fb()
                exchange_and_update()
                return out
            try:
                step = GraphedStep(whole)
            except Exception as exc:      # (every rank runs the same capture: they fail together or not at all)
                # an RCCL build that cannot be captured into a HIP graph: the round-1 structure -- forward + backward as a graph,
                # the collective and the optimizer behind it eagerly -- instead of no line at all
                sys.stderr.write("bench: capturing the whole N > 1 step failed (%s: %s); the collective runs eagerly behind the "
                                 "captured forward + backward\n" % (type(exc).__name__, exc))
                torch.cuda.synchronize()
                fb_g = GraphedStep(fb)

                def step():
                    out = fb_g()
                    exchange_and_update()
                    return out

    def barrier():
        if dp:
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
    if dp:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([n_utt, 1.0], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(cnt)
        total_utt, rccl_ranks = float(cnt[0].item()), int(cnt[1].item())
    else:
        total_utt, rccl_ranks = float(n_utt), 1
    stats = trainer.model._last_ws["stats"].cpu().tolist()
    assert_no_skipped_steps(trainer.model, "timed region (%s, %s)" % (args.module, args.dtype))

    # ---------------------------------------------------------------- roofline: dominant kernel (HIP events) + whole step
    roof = None
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        sb, sf, note = step_work(args.module, params, host_batch, args.dtype == "bf16")
        # the probe runs one extra (eager) training step: on one rank only when there is no collective to join
        dom = dominant_kernel(args.module, params, host_batch, batch, trainer, args.kernel_reps) if world == 1 else None
        tag = "%s_%s_b%d_%s" % (PROFILE_ROUND, args.module, args.batch, args.dtype) if args.module == "cogmen" else \
            "%s_%s" % (PROFILE_ROUND, args.module)      # (the other modules: one counter set per module, at its benched configuration)
        traffic, tsrc = None, None
        # HBM bytes per launch of the dominant kernel from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
        # command (tools/collect_profiles.sh -> tools/pmc_summary.py; gfx950 correction: read bytes = 2 x FETCH_SIZE): one
        # summary per kernel of the step, picked by the dominant kernel's name
        for pmc in sorted(glob.glob(os.path.join(REPO, "profiles", tag + "_*_pmc.json"))):
            with open(pmc) as fh:
                prec = json.load(fh)
            sub = prec.get("kernel_substring", "")
            if dom is not None and sub and sub in dom["kernel"] and prec.get("hbm_bytes_per_launch"):
                traffic, tsrc = prec.get("hbm_bytes_per_launch"), os.path.relpath(pmc, REPO)
        roof = {"bound": None, "kernel": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": traffic,
                "traffic_source": tsrc}
        if dom is not None:
            # the binding roof of the dominant kernel follows from ITS OWN algorithmic intensity: below the ridge of the
            # matrix-core peak its products run on (peak FLOP/s / 8 TB/s) the launch is HBM-side, above it MFMA-side
            tfs, gbs = dom["flops"] / dom["avg_us"] * 1e-6, dom["bytes"] / dom["avg_us"] * 1e-3
            intensity, ridge = dom["flops"] / dom["bytes"], dom["peak"] * 1e12 / (HBM_PEAK_GBS * 1e9)
            hbm_side = intensity < ridge
            roof.update({
                "bound": "hbm" if hbm_side else "mfma", "kernel": dom["kernel"],
                "achieved": gbs if hbm_side else tfs, "peak": HBM_PEAK_GBS if hbm_side else dom["peak"],
                "unit": "GB/s" if hbm_side else "TFLOP/s",
                "frac": gbs / HBM_PEAK_GBS if hbm_side else tfs / dom["peak"],
                "intensity_flop_per_byte": intensity, "ridge_flop_per_byte": ridge,
                "other_roof": {"bound": "mfma" if hbm_side else "hbm", "achieved": tfs if hbm_side else gbs,
                               "peak": dom["peak"] if hbm_side else HBM_PEAK_GBS, "unit": "TFLOP/s" if hbm_side else "GB/s",
                               "frac": tfs / dom["peak"] if hbm_side else gbs / HBM_PEAK_GBS},
                "peak_note": (dom["peak_note"] or "fp32 matrix-core peak (v_mfma_f32_16x16x4_f32): the products of this "
                              "kernel are exact fp32") + "; HBM3E 8 TB/s",
                "algorithmic_flops_per_launch": dom["flops"], "algorithmic_bytes_per_launch": dom["bytes"],
                "traffic_over_algorithmic": traffic / dom["bytes"] if traffic else None,
                "avg_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"],
                "share_of_step": dom["share_us"] / (ms_step * 1e3)})
        roof["step"] = {"algorithmic_bytes": sb, "algorithmic_flops": sf, "work": note, "ms": ms_step,
                        "hbm_GBs": sb / ms_step * 1e-6, "hbm_frac": sb / ms_step * 1e-6 / HBM_PEAK_GBS,
                        "TFLOPs": sf / ms_step * 1e-9, "mfma_f32_frac": sf / ms_step * 1e-9 / MFMA_F32_PEAK_TFS,
                        "mfma_bf16_frac": sf / ms_step * 1e-9 / MFMA_BF16_PEAK_TFS}
        if args.module == "cogmen":
            pr = trainer.model.dominant_kernel_probe(batch, reps=args.kernel_reps)
            ptraffic = None
            ppmc = os.path.join(REPO, "profiles", tag + "_project_graph_pmc.json")
            if os.path.exists(ppmc):
                with open(ppmc) as fh:
                    ptraffic = json.load(fh).get("hbm_bytes_per_launch")
            roof["projection"] = {"bound": "hbm", "kernel": pr["kernel"], "achieved": pr["gbs"], "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": pr["gbs"] / HBM_PEAK_GBS, "traffic": ptraffic,
                                  "algorithmic_bytes": pr["bytes"], "avg_us": pr["us"]}

    # ---------------------------------------------------------------- the same steps, EIGHT per graph launch (rank 0, N=1 only)
    # Between two graph launches the queue idles ~5.5 us (kernel trace: the last kernel of a replay -> the first of the next;
    # inside a graph the kernels follow each other without a gap).  A loop whose next batches are already resident can capture
    # several steps per graph; the headline above stays at one launch per step, as trainer.StepGraphs runs it.
    multi = None
    if rank == 0 and world == 1 and args.module in ("cogmen", "dgcn") and use_graph and not args.no_fp32_path and probe is None:
        S = 8
        step8 = GraphedStep(step_fn, steps=S)
        for _ in range(max(1, args.warmup // S)):
            step8()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(1, args.steps // S)):
            step8()
        torch.cuda.synchronize()
        el8 = time.perf_counter() - t0
        n8 = max(1, args.steps // S) * S
        assert_no_skipped_steps(trainer.model, "multi-step graph")
        multi = {"steps_per_graph": S, "steps": n8, "ms_per_step": 1e3 * el8 / n8, "value": n_utt * n8 / el8, "unit": "utterances/s"}
        del step8

    # ---------------------------------------------------------------- the 1e-4 parity paths, timed the same way (rank 0, N=1 only)
    fp32_path = None
    if rank == 0 and world == 1 and args.module == "cogmen" and args.dtype == "bf16" and not args.no_fp32_path and use_graph \
            and not (args.faithful_dead_encoder or args.chained_encoder):
        # north_star's tolerance (logits within 1e-4 of the reference's fp32 CPU path) is met by the SPLIT compute modes on the
        # same fused 5-launch step (fp32 data, products on the bf16 matrix cores from operands expanded into bf16 terms:
        # tests/test_gpu_cogmen_split.py) and by --compute=f32 (exact-fp32 kernels, 15 launches: tests/test_gpu_cogmen.py).  The
        # headline above is the bf16 compute mode of BASELINE.json configs[1], whose deviation from the unrounded reference is
        # bounded by tests/test_gpu_cogmen.py::test_cogmen_bf16_mode_vs_unrounded_fp32_reference_config2.  Same batch, same step,
        # same HIP-graph replay and the same number of timed steps:
        def timed_path(compute, with_roofline):
            pp = plugin.ParamsType().from_args(["--dataset=" + args.dataset, "--modality=" + args.modality, "--compute=" + compute] + extra)
            pp.train.batch_size = args.batch
            tr = plugin.COGMENTrainer(pp, device)
            bb = tr.prepare_batch(host_batch)
            st = GraphedStep(lambda: tr.train_step(bb))
            for _ in range(args.warmup):
                st()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                st()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            assert_no_skipped_steps(tr.model, "timed region (cogmen, %s)" % compute)
            out = {"dtype": compute, "ms_per_step": 1e3 * el / args.steps, "value": n_utt * args.steps / el, "unit": "utterances/s",
                   "steps": args.steps, "warmup": args.warmup, "loss": tr.model._last_ws["stats"].cpu().tolist()[0],
                   "launches_per_step": None}
            if with_roofline:
                capi.start_recording()
                tr.train_step(bb)
                out["launches_per_step"] = len([e for e in capi.stop_recording()
                                                if not e[0].endswith(("_ok", "_floats", "_doubles", "_workgroup", "_split", "_stamps"))])
                d = dominant_kernel("cogmen", pp, host_batch, bb, tr, args.kernel_reps)
                if d is not None:
                    tfs, gbs = d["flops"] / d["avg_us"] * 1e-6, d["bytes"] / d["avg_us"] * 1e-3
                    hbm_side = d["flops"] / d["bytes"] < d["peak"] * 1e12 / (HBM_PEAK_GBS * 1e9)
                    ptag = "%s_cogmen_b%d_%s" % (PROFILE_ROUND, args.batch, compute)
                    tr_b, tr_src = None, None
                    for pmc in sorted(glob.glob(os.path.join(REPO, "profiles", ptag + "_*_pmc.json"))):
                        with open(pmc) as fh:
                            prec = json.load(fh)
                        if prec.get("kernel_substring", "") and prec["kernel_substring"] in d["kernel"] and prec.get("hbm_bytes_per_launch"):
                            tr_b, tr_src = prec["hbm_bytes_per_launch"], os.path.relpath(pmc, REPO)
                    out["roofline"] = {"bound": "hbm" if hbm_side else "mfma", "kernel": d["kernel"],
                                       "achieved": gbs if hbm_side else tfs, "peak": HBM_PEAK_GBS if hbm_side else d["peak"],
                                       "unit": "GB/s" if hbm_side else "TFLOP/s", "frac": gbs / HBM_PEAK_GBS if hbm_side else tfs / d["peak"],
                                       "traffic": tr_b, "traffic_source": tr_src,
                                       "traffic_over_algorithmic": tr_b / d["bytes"] if tr_b else None, "peak_note": d["peak_note"],
                                       "algorithmic_flops_per_launch": d["flops"], "algorithmic_bytes_per_launch": d["bytes"],
                                       "avg_us": d["avg_us"], "share_of_step": d["share_us"] / (out["ms_per_step"] * 1e3)}
            del st, tr, bb
            return out
        fp32_path = timed_path("f32x32", True)
        fp32_path["tolerance"] = ("logits within 1e-4, every gradient within 2e-3 of its tensor's scale of the UNROUNDED fp32 oracle "
                                  "(tests/test_gpu_cogmen_split.py::test_cogmen_split_config2_shape_parity[f32x32], "
                                  "test_cogmen_split_parity[*-f32x32]; measured at this shape: 2.4e-7 / 4.3e-5)")
        fp32_path["what"] = ("fp32 feature block, fp32 activations and gradients in memory; the fused 5-launch step of the bf16 mode with "
                             "every dense product on v_mfma_f32_16x16x32_bf16 from operands expanded into bf16 terms: three terms in the "
                             "forward products (projection, RGCN, QKVS), two in the backward ones (dH1, dH0, weight gradients)")
        fp32_path["other_1e-4_paths"] = {
            "f32x2": dict(timed_path("f32x2", False), note="two terms everywhere: logits 3.6e-6; gradients 6e-5 once the oracle's backward "
                          "uses the path's own ReLU pattern (a unit within 1e-6 of its kink sits on the other side in about every other "
                          "batch of this size and moves a gradient entry by 3e-3 of its tensor's scale: test_cogmen_split_config2_shape_parity[f32x2])"),
            "f32x3": dict(timed_path("f32x3", False), note="three terms everywhere: logits 2.4e-7, gradients 7.8e-6"),
            "f32": dict(timed_path("f32", False), note="exact-fp32 kernels (v_mfma_f32_16x16x4_f32), unfused 15-launch step: the round-1..3 parity path"),
        }

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = min(16, avail)  # the 1-GPU box's CPU share (16 threads); more threads only slow small ops down
        torch.set_num_threads(cores)
        res = cpu_baseline(args.module, params, host_batch)
        cpu = {"value": res["with_dead_encoder"][0], "unit": "utterances/s", "cores": cores, "kind": "port",
               "sample": "%d full train steps of the same B=%d batch (median), oracle/%s.py with the reference's "
                         "own host-side structure (python graph build%s)"
                         % (res["with_dead_encoder"][1], args.batch, args.module,
                            ", dead Transformer encoder" if args.module == "cogmen" else "")}
        if "without_dead_encoder" in res:
            cpu["value_without_dead_encoder"] = res["without_dead_encoder"][0]

    if probe is not None and rank == 0:
        c = probe.cpu().tolist()
        print("clock probe inside the step: %d cycles in %.2f us -> %.3f GHz" % (c[0], c[1] / 100.0, c[0] / (c[1] * 10.0)),
              file=sys.stderr)
    if rank == 0:
        value = total_utt * args.steps / elapsed
        line = {
            "metric": "utterances/sec training step, COGMEN IEMOCAP-6 atv" if args.module == "cogmen" else
            "utterances/sec training step, %s %s %s" % (args.module, args.dataset, args.modality), "value": value,
            "unit": "utterances/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("COGMEN iemocap-cogmen-6 atv train step (BASELINE.json configs[1]): "
                                    if args.module == "cogmen" else "%s %s train step: " % (args.module, args.dataset)) +
                                   "B=%d dialogues/GPU, T=%d, d_a=%d d_t=%d d_v=%d (D=%d), C=%d, "
                                   "graph build + fwd + CE + bwd + optimizer" % (
                                       args.batch, args.max_len, params.hidden_audio, params.hidden_text,
                                       params.hidden_visual, params.hidden_all, params.n_classes),
                       "utterances_per_step_per_gpu": n_utt, "global_batch_dialogues": args.batch * world,
                       "parallelism": "dp%d" % world, "hip_graph": use_graph,
                       "dp_exchange": ("none" if not dp else (
                           "fused into the optimizer launch (peer-mapped buffers, ERC_DP_P2P=1)" if getattr(trainer.model.flat, "p2p", None) is not None
                           else ("eager after the graph" if os.environ.get("ERC_DP_EAGER", "0") == "1" or not use_graph else "captured in the step graph (RCCL all-reduce)"))),
                       "features_dtype": args.dtype, "loss": stats[0],
                       "faithful_dead_encoder": bool(args.faithful_dead_encoder and args.module == "cogmen"),
                       "chained_encoder": bool(args.chained_encoder and args.module == "cogmen")},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if multi is not None:
            line["multi_step_graph"] = multi
        if fp32_path is not None:
            line["fp32_parity_path"] = fp32_path
            line["config"]["bf16_mode_tolerance"] = ("vs the unrounded fp32 reference at this shape: |dlogit| max < 1e-2, mean < 1.5e-3, "
                                                     "gradients < 0.12 norm-wise (tests/test_gpu_cogmen.py::"
                                                     "test_cogmen_bf16_mode_vs_unrounded_fp32_reference_config2)")
        print(json.dumps(line))
    if dp:
        # Leave without tearing the process group down: the captured step graphs hold RCCL work, and destroying the group under
        # them aborted at interpreter exit once in a while (rc 134 AFTER the line was printed).  Every rank has printed / passed
        # the last barrier; os._exit skips destructors and atexit hooks (the profiling runs of tools/ are single-rank).
        sys.stdout.flush(), sys.stderr.flush()
        barrier()
        os._exit(0)


if __name__ == "__main__":
    main()
