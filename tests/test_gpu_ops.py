"""torch.ops.clairsto.* (csrc/torch_ops.cpp): the hot path as PyTorch custom operators - called directly, checked with
torch.library.opcheck, and held bit-equal to the C-ABI path the Engine uses."""
import numpy as np
import pytest
import torch

from conftest import load_models_npz
from weights_recipe import make_weights

pytestmark = pytest.mark.gpu

OPCHECKS = ("test_schema", "test_autograd_registration", "test_faketensor")


def _packed_from_manifest(cls):
    """packed_weights straight from the golden weights recipe, in manifest order - no nn.Module involved"""
    from clairs_to_amd._lib import model_manifest, CvtCfg
    g = load_models_npz(cls)
    w = make_weights(g["manifest"], seed=g["n_out"])
    if cls.startswith("CvT"):
        cfg = CvtCfg()
        cfg.emb_dim[:], cfg.heads[:], cfg.depth[:], cfg.n_out = (16, 64, 128), (1, 3, 4), (1, 2, 3), g["n_out"]
        man = model_manifest(0, cfg)
        cfg_list = [16, 64, 128, 1, 3, 4, 1, 2, 3, g["n_out"]]
    else:
        man = model_manifest(1, None, g["n_out"])
        cfg_list = None
    flat = np.concatenate([np.asarray(w[k], dtype=np.float32).reshape(-1) for k, n in man])
    assert all(np.asarray(w[k]).size == n for k, n in man)
    return g, torch.from_numpy(flat).cuda(), cfg_list


@pytest.mark.parametrize("cls", ["CvT", "CvT_Indel", "BiGRU_NACGT", "BiGRU_NACGT_Indel"])
def test_model_ops_match_reference_logits(cls):
    g, packed, cfg = _packed_from_manifest(cls)
    x = torch.from_numpy(g["x"]).cuda()
    if cfg is not None:
        out = torch.ops.clairsto.cvt_forward(x, packed, cfg)
    else:
        out = torch.ops.clairsto.bigru_forward(x, packed, g["n_out"])
    assert out.shape == (g["n_out"], x.shape[0], 2) and out.dtype == torch.float32 and out.is_cuda
    np.testing.assert_allclose(out.cpu().numpy(), g["logits"], rtol=0, atol=1e-4)
    # second call: cached handle, same bits; a batch slice is batch-invariant
    again = torch.ops.clairsto.cvt_forward(x, packed, cfg) if cfg is not None else torch.ops.clairsto.bigru_forward(x, packed, g["n_out"])
    assert torch.equal(out, again)
    # in-place weight update bumps the tensor version: the operator must rebuild its handle
    packed.mul_(0.5)
    changed = torch.ops.clairsto.cvt_forward(x, packed, cfg) if cfg is not None else torch.ops.clairsto.bigru_forward(x, packed, g["n_out"])
    assert not torch.equal(out, changed)
    packed.mul_(2.0)
    back = torch.ops.clairsto.cvt_forward(x, packed, cfg) if cfg is not None else torch.ops.clairsto.bigru_forward(x, packed, g["n_out"])
    assert torch.equal(out, back)


def test_model_ops_reject_bad_arguments():
    g, packed, cfg = _packed_from_manifest("CvT")
    x = torch.from_numpy(g["x"]).cuda()
    with pytest.raises(RuntimeError):
        torch.ops.clairsto.cvt_forward(x[:, :32], packed, cfg)                 # wrong window
    with pytest.raises(RuntimeError):
        torch.ops.clairsto.cvt_forward(x, packed[:-1].contiguous(), cfg)       # manifest size mismatch (CTO_EMISSING)
    with pytest.raises(RuntimeError):
        torch.ops.clairsto.cvt_forward(x, packed, cfg[:9])
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.clairsto.cvt_forward(x.cpu(), packed.cpu(), cfg)             # no CPU kernel: there is no CPU fallback
    empty = torch.ops.clairsto.cvt_forward(x[:0], packed, cfg)
    assert empty.shape == (4, 0, 2)


def test_ops_path_is_bit_equal_to_the_c_abi_path():
    """featurize -> AFF / NEG -> posterior through torch.ops vs the Engine (ctypes on the C ABI): same kernels, same bits"""
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.featurize import featurize_op
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    dev = torch.device("cuda:0")
    chunk = SynthChunk(300, seed=3)
    models = synthetic_models(4)
    lik, edges = lik_and_edges(likelihood_table(4), 4)
    eng = Engine(models["aff"], models["neg"], lik, edges, min_bq=20, device=dev)
    dp = eng.upload(chunk.arrays())
    sp = torch.from_numpy(chunk.site_pos).to(dev)
    want = eng.run_device(dp, sp)
    feat = featurize_op(dp, sp, 20, 50)
    assert torch.equal(feat.x_aff, want["features"].x_aff) and torch.equal(feat.x_neg, want["features"].x_neg)
    assert torch.equal(feat.site_info, want["site_info"])
    centre = feat.site_info[:, 0].long()          # the op returns every column's vector, the engine the candidate columns' only
    assert torch.equal(feat.colvec.index_select(0, centre), want["features"].site_colvec)
    la = torch.stack(models["aff"].to(dev)(feat.x_aff))         # nn.Module shims: forward goes through torch.ops
    ln = torch.stack(models["neg"].to(dev)(feat.x_neg))
    assert torch.equal(la, want["aff_logits"]) and torch.equal(ln, want["neg_logits"])
    probs, post, dec, qual = torch.ops.clairsto.posterior(la, ln, torch.from_numpy(lik).to(dev), torch.from_numpy(edges).to(dev))
    assert torch.equal(probs, want["probs"]) and torch.equal(post, want["post"]) and torch.equal(dec, want["decision"])
    assert torch.equal(qual, want["qual"])


def test_opcheck():
    """torch.library.opcheck: schema correctness, autograd registration, FakeTensor (Meta kernel) agreement"""
    from torch.library import opcheck
    from clairs_to_amd.engine import synthetic_models
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    from clairs_to_amd.pack import DevicePack
    dev = torch.device("cuda:0")
    g, packed, cfg = _packed_from_manifest("CvT")
    x = torch.from_numpy(g["x"]).cuda()
    opcheck(torch.ops.clairsto.cvt_forward.default, (x, packed, cfg), test_utils=OPCHECKS)
    g2, packed2, _ = _packed_from_manifest("BiGRU_NACGT_Indel")
    opcheck(torch.ops.clairsto.bigru_forward.default, (torch.from_numpy(g2["x"]).cuda(), packed2, 6), test_utils=OPCHECKS)
    lik, edges = lik_and_edges(likelihood_table(4), 4)
    la = torch.randn(4, 64, 2, device=dev)
    ln = torch.randn(4, 64, 2, device=dev)
    opcheck(torch.ops.clairsto.posterior.default, (la, ln, torch.from_numpy(lik).to(dev), torch.from_numpy(edges).to(dev)),
            test_utils=OPCHECKS)
    opcheck(torch.ops.clairsto.softmax2.default, (la,), test_utils=OPCHECKS)
    chunk = SynthChunk(40, seed=8)
    dp = DevicePack(chunk.arrays(), dev)
    t = dp.t
    sp = torch.from_numpy(chunk.site_pos).to(dev)
    opcheck(torch.ops.clairsto.pileup_featurize.default,
            (t["entries"], t["col_off"], t["col_pos"], t["col_ref"], t["key_off"], t["key_meta"], t["key_group"], sp, 20, 50),
            test_utils=OPCHECKS)


def test_ops_trace_under_fake_tensors():
    """the Meta kernels give the output shapes the real kernels produce (what torch.compile's tracer needs)"""
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(17, 33, 34, device="cuda")
        w = torch.empty(1000, device="cuda")
        assert torch.ops.clairsto.cvt_forward(x, w, [16, 64, 128, 1, 3, 4, 1, 2, 3, 6]).shape == (6, 17, 2)
        assert torch.ops.clairsto.bigru_forward(x, w, 4).shape == (4, 17, 2)
        lo = torch.empty(4, 17, 2, device="cuda")
        p, post, dec, q = torch.ops.clairsto.posterior(lo, lo, torch.empty(4, 10, 10, device="cuda", dtype=torch.float64),
                                                       torch.empty(8, 11, device="cuda", dtype=torch.float64))
        assert p.shape == (17, 8, 2) and post.shape == (17, 4) and dec.dtype == torch.int32 and q.dtype == torch.float64


def test_apply_softmax_modules_return_probabilities_from_the_device_op():
    """a module built with apply_softmax=True (clairs/model.py:255-259, :461-465) returns Softmax(dim=1) of every head: computed by
    clairsto::softmax2 (cto_softmax_pairs), the arithmetic of the epilogue's own softmax, incl. saturated and equal logits"""
    from clairs_to_amd.nn_shims import from_state_dict
    dev = torch.device("cuda:0")
    lg = torch.tensor([[[-30.0, 30.0], [30.0, -30.0], [0.0, 0.0], [1e-3, -1e-3], [-1.7, 20.0]]], device=dev)
    got = torch.ops.clairsto.softmax2(lg).cpu().numpy()
    want = torch.softmax(lg.cpu().double(), dim=-1).numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-7)
    assert got[0, 0, 1] == 1.0 and 0.0 < got[0, 1, 1] < 1e-25 and got[0, 2, 0] == 0.5
    assert torch.ops.clairsto.softmax2(torch.empty((4, 0, 2), device=dev)).shape == (4, 0, 2)
    for cls in ("CvT", "BiGRU_NACGT_Indel"):
        g = load_models_npz(cls)
        m = from_state_dict(cls, make_weights(g["manifest"], seed=g["n_out"]))
        x = torch.from_numpy(g["x"][:40]).to(dev)
        plain = torch.stack(m(x))
        m.apply_softmax = True
        probs = torch.stack(m(x))
        np.testing.assert_array_equal(probs.cpu().numpy(), torch.ops.clairsto.softmax2(plain).cpu().numpy())
        p_ref = torch.softmax(torch.from_numpy(g["logits"][:, :40]), dim=-1).numpy()
        assert np.abs(probs.cpu().numpy() - p_ref).max() < 1e-4
