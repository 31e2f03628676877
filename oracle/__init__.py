"""CPU oracle for the ERC conversation-graph training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``erc_amd`` + ``libercgraft.so``) never imports this package
and raises if the HIP library is missing.

What it is: a plain PyTorch-CPU fp32 restatement of the reference's hot path
(sailist/emotion-recognition-in-conversation, ``track_mm/{cogmen,dagerc,mmgcn,
dgcn}.py`` and their ``*_models.py`` / ``*_utils.py``), structurally faithful
to the reference including its host-side per-edge Python graph construction,
so that it doubles as the "reference CPU path" timed next to the GPU number.
Every function cites the reference file:line it follows.

Pinning status (see DESIGN.md "Oracle"):
  * graph construction (window edges / relation ids / DAG adjacency / speaker
    mask), ERCCollate batch layout, DAG-ERC end-to-end logits+grads, MMGCN
    end-to-end logits+grads, DialogueGCN EdgeAtt / batch_graphify / vendored
    RGCNConv / SeqContext / Classifier, the vendored encoder layer
    (contrib/nn.py:206-305, two layers, with and without key-padding mask):
    PINNED by golden vectors produced by importing the reference's own modules
    in the build container (``tests/golden/make_golden.py`` -- one ``gen_*``
    per item above -- fixtures committed as ``.npz``; the CPU tests
    ``tests/test_oracle_*.py`` / ``test_host_layout.py`` hold the oracle to
    them).
  * torch_geometric ``RGCNConv`` (mean), ``TransformerConv`` (heads=1) and
    ``GraphConv`` are third-party, unpinned (``requirements.txt:12``) and not
    installed: restated here from their published formulae ->
    "parity unpinned" for those three operators.
  * COGMEN's bf16 compute mode is checked against the SAME restatement with the
    operands of the products that mode runs on bf16 matrix cores rounded to
    bf16 (``pyg.RoundedLinear`` / ``pyg.RGCNMeanRounded``; their hand-written
    backward formulas equal autograd when the rounding is switched off:
    ``tests/test_oracle_cogmen_rounded.py``) -- the reference has no bf16 path,
    so this pins implementation error of the mode, not the mode's quantisation.
"""
