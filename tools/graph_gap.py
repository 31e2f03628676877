#!/usr/bin/env python3
"""How much of a COGMEN step is the bubble BETWEEN two graph launches?  Times K steps replayed as K one-step graphs against
K / S replays of an S-step graph, and the host cost of a replay call."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from bench import synthetic_batch
import track_mm.cogmen as plugin

params = plugin.ParamsType().from_args(["--dataset=iemocap-cogmen-sbert-6", "--modality=atv", "--compute=bf16"])
tr = plugin.COGMENTrainer(params, "cuda:0")
batch = tr.prepare_batch(synthetic_batch(params, 32, 110, seed=1))
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
K = 2400
for S in (1, 2, 4, 8):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(S):
            tr.train_step(batch)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K // S):
        g.replay()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("steps per graph %d: %.2f us per step (host enqueue %.2f us per replay)" % (S, t / K * 1e6, t_host / (K // S) * 1e6))

# two graph execs of the same step, replayed alternately (does the runtime serialize launches of ONE exec harder?)
gs = []
for _ in range(2):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        tr.train_step(batch)
    gs.append(g)
for _ in range(20):
    gs[0].replay(), gs[1].replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K // 2):
    gs[0].replay()
    gs[1].replay()
torch.cuda.synchronize()
print("two execs alternating: %.2f us per step" % ((time.perf_counter() - t0) / K * 1e6))

# the same one-step graph launched alternately on two streams, ordered by events (is the bubble a property of one queue?)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
g1 = gs[0]
evs = [torch.cuda.Event(), torch.cuda.Event()]
torch.cuda.synchronize()
def run_alt(n):
    prev = None
    for i in range(n):
        st = s0 if i % 2 == 0 else s1
        with torch.cuda.stream(st):
            if prev is not None:
                st.wait_event(prev)
            g1.replay()
            ev = evs[i % 2]
            ev.record(st)
            prev = ev
run_alt(40)
torch.cuda.synchronize()
t0 = time.perf_counter()
run_alt(K)
torch.cuda.synchronize()
print("one exec, two streams alternating with events: %.2f us per step" % ((time.perf_counter() - t0) / K * 1e6))
