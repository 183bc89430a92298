"""Repository contract checks (CPU): the oracle is test infrastructure only, the product has no CPU fallback."""
import os
import re

from conftest import ROOT

PKG = os.path.join(ROOT, "clairs_to_amd")


def _product_sources():
    for d, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")) and "build" not in d:
                yield os.path.join(d, f)


def test_product_never_touches_the_oracle_or_the_reference():
    bad = []
    for path in _product_sources():
        src = open(path).read()
        if re.search(r"^\s*(import|from)\s+oracle\b", src, re.M) or "cto_oracle" in src or "liboracle" in src:
            bad.append(path)
        if "/root/reference" in src:
            bad.append(path)
    assert not bad, bad


def test_oracle_header_says_test_infrastructure():
    head = open(os.path.join(ROOT, "oracle", "cto_oracle.c")).read(1200)
    assert "TEST INFRASTRUCTURE ONLY" in head


def test_product_fails_loudly_without_gpu():
    import pytest
    import torch
    from clairs_to_amd.pack import DevicePack
    from clairs_to_amd.engine import Engine
    with pytest.raises(RuntimeError):
        DevicePack(dict(col_pos=[], col_ref=[], col_off=[0], key_off=[0], entries=[], key_meta=[]), "cpu")
    with pytest.raises(RuntimeError):
        Engine(None, None, None, None, device="cpu")


def test_required_files_exist():
    for f in ("bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "include/clairsto_amd.h",
              "oracle/cto_oracle.c", "tests/golden/gen_golden.py"):
        assert os.path.exists(os.path.join(ROOT, f)), f
