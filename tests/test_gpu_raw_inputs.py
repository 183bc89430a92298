"""The networks on the un-rescaled int16 tensors (cto_model_forward_raw; csrc/gru_kernel.h XRAW, csrc/cvt_block.h phase 0): the first
layers rescale where they load - float(double(v) * min_rescale_cov / depth), the expression of clairs/predict.py:172-207 as the tensor
kernel states it - so the logits must equal the fp32 hand-over's bit for bit, and the fp32 tensors need not exist."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _engine(K, dev, **kw):
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.synth import likelihood_table, lik_and_edges
    models = synthetic_models(K)
    lik, edges = lik_and_edges(likelihood_table(K), K)
    return Engine(models["aff"], models["neg"], lik, edges, device=dev, **kw)


def _logits(handle, raw, feat, which, cov, B, K, dev):
    import torch
    from clairs_to_amd._lib import lib, check
    out = torch.empty((K, B, 2), device=dev)
    s = int(torch.cuda.current_stream().cuda_stream)
    if raw:
        x = feat.raw_aff if which == 0 else feat.raw_neg
        check(lib.cto_model_forward_raw(handle, x.data_ptr(), feat.site_info.data_ptr(), which, cov, B, out.data_ptr(), s))
    else:
        x = feat.x_aff if which == 0 else feat.x_neg
        check(lib.cto_model_forward(handle, x.data_ptr(), B, out.data_ptr(), s))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("K", [4, 6])
def test_raw_inputs_give_the_fp32_hand_overs_logits_bit_for_bit(dev, K):
    """both networks, a full 4096-site chunk and ragged batches (whole and partial 32- / 16-site tiles of the recurrent kernel, partial
    CvT tiles), rescale on (mean depth 70: most sites are deeper than 50) and off, and the tensors the record derives on demand equal the ones
    the tensor kernel writes"""
    import torch
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk
    eng = _engine(K, dev)
    chunk = SynthChunk(4096, seed=31, depth_mean=70.0)
    dp = eng.upload(chunk.arrays())
    sp = torch.from_numpy(chunk.site_pos).to(dev)
    for cov in (50, 0):
        both = featurize(dp, sp, 20, cov, want_raw=True, want_x=True)
        only_raw = featurize(dp, sp, 20, cov, want_raw=True, want_x=False)
        assert only_raw._x == [None, None]
        assert torch.equal(only_raw.x_aff, both.x_aff) and torch.equal(only_raw.x_neg, both.x_neg)      # derived on first use
        if cov:
            assert float((both.site_info[:, 1] > cov).float().mean()) > 0.5
        for handle, which in ((eng.h_aff, 0), (eng.h_neg, 1), (eng.h_neg, 0)):
            for B in (4096, 1, 15, 16, 17, 33, 1000):
                want = _logits(handle, False, both, which, cov, B, K, dev)
                got = _logits(handle, True, only_raw, which, cov, B, K, dev)
                np.testing.assert_array_equal(got, want, err_msg="which %d B %d cov %d" % (which, B, cov))


def test_engine_on_raw_inputs_equals_the_fp32_engine(dev):
    import torch
    from clairs_to_amd.synth import SynthChunk
    chunk = SynthChunk(1500, seed=7)
    e_raw, e_f32 = _engine(4, dev, raw_inputs=True), _engine(4, dev, raw_inputs=False)
    assert e_raw.raw_inputs and not e_f32.raw_inputs
    assert _engine(4, dev).raw_inputs == (os.environ.get("CTO_RAW_INPUTS", "0") == "1")      # the fp32 hand-over is the default (engine.py)
    a = e_raw.run_chunk(chunk.arrays(), chunk.site_pos)
    b = e_f32.run_chunk(chunk.arrays(), chunk.site_pos)
    torch.cuda.synchronize()
    for k in ("aff_logits", "neg_logits", "probs", "post", "decision", "qual"):
        assert torch.equal(a[k], b[k]), k
    assert a["features"]._x == [None, None] and a["features"].raw_aff is not None          # no fp32 tensor was written
    assert torch.equal(a["features"].x_aff, b["features"].x_aff)
    # Illumina: the NEG network reads the AFF pass
    e1, e2 = _engine(4, dev, neg_reads_aff=True, raw_inputs=True), _engine(4, dev, neg_reads_aff=True, raw_inputs=False)
    a, b = e1.run_chunk(chunk.arrays(), chunk.site_pos), e2.run_chunk(chunk.arrays(), chunk.site_pos)
    torch.cuda.synchronize()
    assert torch.equal(a["neg_logits"], b["neg_logits"]) and torch.equal(a["probs"], b["probs"])


def test_handles_without_an_int16_loader_expand_inside_the_call(dev, monkeypatch):
    """split-operand handles (their first layers read fp32): cto_model_forward_raw expands the tensor once and gives what
    cto_model_forward gives on the fp32 tensor"""
    import torch
    from clairs_to_amd.engine import Engine, synthetic_models
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.synth import SynthChunk, likelihood_table, lik_and_edges
    models = synthetic_models(4)
    for m in (models["aff"], models["neg"]):
        m.split_operands = "f16"
    lik, edges = lik_and_edges(likelihood_table(4), 4)
    eng = Engine(models["aff"], models["neg"], lik, edges, device=dev)
    chunk = SynthChunk(700, seed=3, depth_mean=70.0)
    dp, sp = eng.upload(chunk.arrays()), torch.from_numpy(chunk.site_pos).to(dev)
    both = featurize(dp, sp, 20, 50, want_raw=True, want_x=True)
    for handle, which in ((eng.h_aff, 0), (eng.h_neg, 1)):
        np.testing.assert_array_equal(_logits(handle, True, both, which, 50, 700, 4, dev), _logits(handle, False, both, which, 50, 700, 4, dev))
