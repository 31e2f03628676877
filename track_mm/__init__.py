"""Method plugins of the ERC path: one module per ``--module=`` value, each exposing a
zero-argument ``main()`` (contract of train_mm.py:13-25 of the reference)."""
