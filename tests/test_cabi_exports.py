"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/ercgraft.h declares.
No compute call is made here (no GPU in the build container)."""
import os
import re

from erc_amd import capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    capi.build()
    lib = capi.lib()
    header = open(os.path.join(REPO, "include", "ercgraft.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(erc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.erc_abi_version() == 3


def test_argument_errors_are_reported_not_thrown():
    lib = capi.lib()
    rc = lib.erc_slab_reduce(None, 1, 0, None, 0, 0, None, 0, 0, None)
    assert rc == -1 and b"slab_reduce" in lib.erc_last_error()


def test_product_never_imports_the_oracle():
    pkg = capi._HERE
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(root, f)).read()
                assert "oracle" not in re.sub(r"#.*", "", text).replace("the oracle", ""), f
    for f in ("train_mm.py",):
        path = os.path.join(REPO, f)
        if os.path.exists(path):
            assert "import oracle" not in open(path).read()
