import base64
import gzip
import hashlib
import io
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the built libraries are git-ignored: a fresh checkout builds them before the first test imports the package
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "clairs_to_amd", "libclairsto_amd.so")):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "clairs_to_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def load_json_gz(name):
    with gzip.open(os.path.join(GOLDEN, name), "rt") as f:
        return json.load(f)


def load_models_npz(cls):
    z = np.load(os.path.join(GOLDEN, "models_%s.npz" % cls))
    manifest = [(k, tuple(s)) for k, s in json.loads(str(z["manifest"]))]
    return dict(x=z["x"], logits=z["logits"], manifest=manifest, n_out=int(z["n_out"]))


def parse_tensor_text(text):
    """rows of the reference's tensor text -> (ctg,pos,ref_seq) list, int32 [n,33,34], alt_info list"""
    rows = [r.split("\t") for r in text.strip().split("\n") if r]
    X = np.array([[int(v) for v in r[3].split()] for r in rows], dtype=np.int32).reshape(-1, 33, 34)
    return rows, X, [r[4] for r in rows]


@pytest.fixture(scope="session")
def golden_region():
    return load_json_gz("region.json.gz")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(autouse=True)
def _poisoned_lds(request):
    """Every GPU test starts on LDS full of NaN patterns (cto_debug_poison_lds): results must not depend on what an
    earlier kernel left in shared memory."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            from clairs_to_amd._lib import lib, check, current_stream_ptr
            check(lib.cto_debug_poison_lds(current_stream_ptr()))
    yield


@pytest.fixture(scope="session")
def region2k():
    """fixture + the inputs regenerated from its seed (checked against the stored SHA-256: a drifting generator must fail
    here, not look like a parity failure)"""
    from clairs_to_amd.synth import SynthChunk, mpileup_text
    g = load_json_gz("region2k.json.gz")
    kw = dict(g["params"])
    chunk = SynthChunk(kw.pop("n_sites"), **kw)
    ref, ref_lo = chunk.ref_window()
    texts = {"neg": mpileup_text(chunk, min_bq=0), "aff": mpileup_text(chunk, min_bq=g["min_bq_aff"])}
    sha = lambda t: hashlib.sha256(t.encode()).hexdigest()
    assert sha(texts["neg"]) == g["input_sha"]["mpileup_neg"] and sha(texts["aff"]) == g["input_sha"]["mpileup_aff"]
    assert sha(ref) == g["input_sha"]["ref"]
    return dict(g=g, chunk=chunk, ref=ref, ref_lo=ref_lo, texts=texts, sites=chunk.site_pos.tolist())


def load_genuine_pickle(name):
    """torch.load of a pickle the REFERENCE's classes wrote, resolved onto the shims (clairs.model aliases)"""
    import torch
    from clairs_to_amd import nn_shims
    nn_shims.install_reference_aliases()
    g = load_json_gz("pickles.json.gz")[name]
    raw = gzip.decompress(base64.b64decode(g["pickle_gz_b64"]))
    return torch.load(io.BytesIO(raw), map_location="cpu", weights_only=False)[g["key"]], g


def load_range_npz(cls):
    """models_<cls>_range.npz (tests/golden/gen_range.py): the reference modules outside the O(1) regime, fp32 and fp64."""
    z = np.load(os.path.join(GOLDEN, "models_%s_range.npz" % cls))
    manifest = [(k, tuple(s)) for k, s in json.loads(str(z["manifest"]))]
    sets = [(n, float(s), float(g)) for n, s, g in json.loads(str(z["sets"]))]
    return dict(x=z["x"], manifest=manifest, n_out=int(z["n_out"]), sets=sets, z=z)


def range_errors(logits, z, name):
    """(max |dP| vs the reference's fp64 run, vs its fp32 run, relative logit error vs fp64, the fp32 reference's own
    |dP| vs fp64, its own relative logit error) for one range set."""
    l64 = z["logits64_" + name]
    p64, p32 = z["probs64_" + name], z["probs32_" + name]
    lg = np.asarray(logits, dtype=np.float64)
    e = np.exp(lg - lg.max(-1, keepdims=True))
    p = e / e.sum(-1, keepdims=True)
    scale = max(1.0, float(np.abs(l64).max()))
    return dict(dp64=float(np.abs(p - p64).max()), dp32=float(np.abs(p - p32.astype(np.float64)).max()),
                rel=float(np.abs(lg - l64).max() / scale),
                ref_dp=float(np.abs(p32.astype(np.float64) - p64).max()),
                ref_rel=float(np.abs(z["logits32_" + name].astype(np.float64) - l64).max() / scale))
