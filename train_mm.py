#!/usr/bin/env python3
"""Plugin dispatcher with the reference's command line (train_mm.py:13-25):

    python3 train_mm.py --module=cogmen --dataset=iemocap-cogmen-6 --modality=atv [--key=value ...]

``--module`` names a module of the ``track_mm`` package that exposes ``main()``;
an unknown / missing module prints the available set and exits 1.
"""
import importlib
import sys
from pkgutil import iter_modules
from pprint import pprint

import track_mm as track

methods = {m.name for m in iter_modules(track.__path__)}

if __name__ == "__main__":
    module, rest = None, []
    for tok in sys.argv[1:]:
        if tok.startswith("--module="):
            module = tok.split("=", 1)[1]
        else:
            rest.append(tok)
    if module is None or module not in methods:
        print("--module=")
        pprint(methods)
        sys.exit(1)
    sys.argv = [sys.argv[0]] + rest + ["--module=" + module]
    importlib.import_module("%s.%s" % (track.__name__, module)).main()
