#!/usr/bin/env python3
"""Where a DAG-ERC recurrence step spends its time: runs the bench workload's forward / backward recurrence of one layer
with phase stamps switched on (erc_dag_rec_set_stamps) and prints the average per-phase time of workgroup 0.

    python tools/dag_stamps.py [--epc E --dg D]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epc", type=int, default=0), ap.add_argument("--dg", type=int, default=0)
    ap.add_argument("--lpl", type=int, default=0)
    ap.add_argument("--bepc", type=int, default=0), ap.add_argument("--bdg", type=int, default=0)
    ap.add_argument("--blpl", type=int, default=0)
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    for k, v in (("ERC_DAG_EPC", a.epc), ("ERC_DAG_DG", a.dg), ("ERC_DAG_LPL", a.lpl), ("ERC_DAG_BEPC", a.bepc), ("ERC_DAG_BDG", a.bdg), ("ERC_DAG_BLPL", a.blpl)):
        if v:
            os.environ[k] = str(v)
    from bench import synthetic_batch
    from erc_amd import capi
    import track_mm.dagerc as plugin
    params = plugin.ParamsType().from_args(["--dataset=iemocap-cogmen-6", "--modality=atv", "--reimplement"])
    tr = plugin.DAGERCTrainer(params, "cuda:0")
    batch = tr.prepare_batch(synthetic_batch(params, a.batch, 110, seed=1))
    for _ in range(3):
        tr.train_step(batch)
    T = 110
    names = {"erc_dag_rec_fwd": ["start", "polled M", "gates done", "barrier A", "polled h | h published", "R done | saves done",
                                 "barrier B", "tail done (EW)"],     # workgroup 0 = layer 0, slice 0
             "erc_dag_rec_bwd": ["start", "E1 done", "M1+publish done", "partials summed", "E2 | Y done", "barrier 3", "dots done",
                                 "E3 done (EW)"]}      # workgroup 0 = the launch's lowest layer, slice 0
    capi.start_recording()
    tr.train_step(batch)
    rec = capi.stop_recording()
    torch.cuda.synchronize()
    print("config (epc, dg, groups per launch, layers per launch) forward | backward:", tr.model._last_ws["cfg"])
    for entry in ("erc_dag_rec_fwd", "erc_dag_rec_bwd"):
        call = [e for e in rec if e[0] == entry][0]
        st = torch.zeros(T, 2, 8, dtype=torch.int64, device="cuda:0")
        capi.dag_rec_set_stamps(st)
        capi.replay(call)
        torch.cuda.synchronize()
        capi.dag_rec_set_stamps(None)
        s = st.cpu().double()
        steps = list(range(30, 90))
        print("==", entry, "(cycles of the 100 MHz-independent shader clock; average over steps 30..89)")
        for role, rname in ((0, "matrix wavefront 0"), (1, "elementwise wavefront")):
            base = s[steps, role, 0]
            line = []
            for k in range(1, 8):
                v = s[steps, role, k]
                ok = v > 0
                if ok.any():
                    line.append("%s: +%.0f" % (names[entry][k], float((v - base)[ok].mean())))
            print("  %-22s %s" % (rname, " | ".join(line)))
        d = s[steps[1:], 0, 0] - s[steps[:-1], 0, 0]
        print("  step period: %.0f cycles (abs)" % float(d.abs().mean()))


if __name__ == "__main__":
    main()
