"""End-to-end hot path for one GPU: column pack -> pileup tensors -> AFF + NEG networks -> posterior.

This is what replaces STEP 2a-2d / 6-1..6-3 of run_clairs_to (reference: run_clairs_to:1228-1308, 1562-1647)
for one chunk of candidate sites.  Everything between the pack upload and the posterior stays in HBM:
no text tensors, no gzip pipes, no host round trips.  The AFF and NEG networks are independent given the two
tensors, so they run on two HIP streams.
"""
import numpy as np
import os

import torch

from ._lib import lib, check, c_vp
from .call_variants import Posterior
from .featurize import featurize
from .nn_shims import CvT, CvT_Indel, BiGRU_NACGT, BiGRU_NACGT_Indel
from .pack import DevicePack

CVT_PREDICT_CFG = dict(s1_emb_dim=16, s2_emb_dim=64, s3_emb_dim=128, s1_heads=1, s2_heads=3, s3_heads=4,
                       s1_depth=1, s2_depth=2, s3_depth=3)        # clairs/predict.py:520-553


def random_state_dict(module, seed=0, head_gain=2.0):
    """Seeded synthetic weights (no pretrained weights exist offline): uniform with fan-in scaling so that
    activations stay O(1); BatchNorm statistics randomised.  Returns dict name -> float32 numpy array."""
    import zlib
    out = {}
    for name, t in module.state_dict().items():
        if name.endswith("num_batches_tracked"):
            continue
        shape = tuple(t.shape)
        r = np.random.default_rng([zlib.crc32(name.encode()) & 0xffffffff, seed])
        if name.endswith("running_var"):
            a = r.uniform(0.5, 1.5, size=shape)
        elif name.endswith("running_mean"):
            a = r.uniform(-0.2, 0.2, size=shape)
        elif name.endswith(".g") or ".net.1.weight" in name:
            a = r.uniform(0.8, 1.2, size=shape)
        elif name.endswith(".b") or name.endswith("bias") or "bias_" in name:
            a = r.uniform(-0.1, 0.1, size=shape)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            if len(shape) == 4 and shape[1] == 1:
                fan_in = 3
            elif len(shape) == 4 and shape[2] == 3:
                fan_in = shape[1] * 3
            bound = (3.0 / fan_in) ** 0.5
            if "_fc3" in name:
                bound *= head_gain
            if name.startswith("layer1.0.") or name.startswith("lstm.weight_ih"):
                bound *= 0.05
            a = r.uniform(-bound, bound, size=shape)
        out[name] = a.astype(np.float32)
    return out


CVT_CONSTRUCTOR_CFG = dict(s1_emb_dim=32, s2_emb_dim=64, s3_emb_dim=128, s1_heads=1, s2_heads=3, s3_heads=6,
                           s1_depth=1, s2_depth=2, s3_depth=10)   # clairs/model.py:153-184 (what a pickled SNV module may carry)


def synthetic_models(n_out=4, seed=0, cvt_cfg=None):
    """Random-init AFF/NEG modules of the reference architecture (predict.py configuration unless cvt_cfg says otherwise)."""
    aff_cls, neg_cls = (CvT, BiGRU_NACGT) if n_out == 4 else (CvT_Indel, BiGRU_NACGT_Indel)
    aff = aff_cls(model_type="acgt", **(CVT_PREDICT_CFG if cvt_cfg is None else cvt_cfg)).eval()
    neg = neg_cls(model_type="nacgt").eval()
    out = {}
    for tag, m in (("aff", aff), ("neg", neg)):
        w = random_state_dict(m, seed=seed + n_out)
        sd = m.state_dict()
        for k, v in w.items():
            sd[k] = torch.from_numpy(v)
        m.load_state_dict(sd)
        out[tag] = m
        out[tag + "_weights"] = w
    return out


class Engine:
    """One GPU's worth of the hot path.  aff/neg: nn_shims modules (or anything exposing `_handle()`)."""

    def __init__(self, aff, neg, lik, edges, min_bq=20, min_rescale_cov=50, device="cuda", two_streams=False,
                 neg_reads_aff=False, raw_inputs=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("clairs_to_amd.Engine needs a HIP device; there is no CPU fallback")
        self.aff, self.neg = aff, neg
        self.K = len(aff._heads_out)
        assert len(neg._heads_out) == self.K
        self.min_bq, self.min_rescale_cov = int(min_bq), int(min_rescale_cov)
        # Illumina: the NEG tensor files are symlinks to the AFF ones (run_clairs_to:1248-1252) - the NEG network reads the
        # --min_bq pass, not the BQ >= 0 pass.  Identical for the platform default (min_bq 0); differs with an explicit --min_bq.
        self.neg_reads_aff = bool(neg_reads_aff)
        # raw_inputs: the networks read the int16 tensors and rescale where they load them (cto_model_forward_raw) - the fp32 tensors, 18 MB
        # per network and 4096-site step, are then never written.  Same logits bit for bit, but measured no faster on MI355X (BiGRU layer
        # 1 0.278 -> 0.284 ms - its register budget, not the f64 arithmetic -, the CvT unchanged, a step 1.874 -> 1.879 ms: A/B pairs), so the fp32 hand-over stays the default; CTO_RAW_INPUTS=1 / raw_inputs=True select the int16 one.
        self.raw_inputs = (os.environ.get("CTO_RAW_INPUTS", "0") == "1") if raw_inputs is None else bool(raw_inputs)
        with torch.cuda.device(self.device):
            self.h_aff, self.h_neg = aff._handle(), neg._handle()
            self.posterior = Posterior(lik, edges, self.device)
            self.s_neg = torch.cuda.Stream(self.device) if two_streams else None
        self.macs_per_site = int(lib.cto_model_macs_per_site(self.h_aff)) + int(lib.cto_model_macs_per_site(self.h_neg))

    def upload(self, arrays):
        return DevicePack(arrays, self.device)

    def run_device(self, dev_pack, site_pos_dev, want_raw=False):
        """All inputs already resident in HBM. Returns device tensors."""
        with torch.cuda.device(self.device):
            raw = self.raw_inputs
            feat = featurize(dev_pack, site_pos_dev, self.min_bq, self.min_rescale_cov, want_raw=want_raw or raw, want_x=not raw)
            B, K = site_pos_dev.numel(), self.K
            la = torch.empty((K, B, 2), dtype=torch.float32, device=self.device)
            ln = torch.empty((K, B, 2), dtype=torch.float32, device=self.device)
            main = torch.cuda.current_stream()
            cov = int(self.min_rescale_cov) if self.min_rescale_cov else 0
            neg_pass = 0 if self.neg_reads_aff else 1
            if not raw and self.neg_reads_aff:
                feat.x_neg = feat.x_aff

            def forward(handle, which, logits, stream):
                if raw:
                    x = feat.raw_aff if which == 0 else feat.raw_neg
                    check(lib.cto_model_forward_raw(handle, x.data_ptr(), feat.site_info.data_ptr(), which, cov, B, logits.data_ptr(), int(stream.cuda_stream)))
                else:
                    x = feat.x_aff if which == 0 else feat.x_neg
                    check(lib.cto_model_forward(handle, x.data_ptr(), B, logits.data_ptr(), int(stream.cuda_stream)))
                return x

            if self.s_neg is not None:
                self.s_neg.wait_stream(main)
                xn = forward(self.h_neg, neg_pass, ln, self.s_neg)
                forward(self.h_aff, 0, la, main)
                main.wait_stream(self.s_neg)
                xn.record_stream(self.s_neg)
                ln.record_stream(self.s_neg)
            else:
                forward(self.h_neg, neg_pass, ln, main)
                forward(self.h_aff, 0, la, main)
            out = self.posterior(la, ln)
        out.update(aff_logits=la, neg_logits=ln, site_info=feat.site_info, features=feat)
        return out

    def run_stream(self, chunks, depth=2):
        """Host buffers in, host results out, uploads hidden behind compute: `chunks` yields (arrays, site_pos) with `arrays`
        either numpy arrays or the pinned tensors of `pack.pin_arrays`; the pack of chunk i+1 crosses PCIe on a copy stream
        while chunk i computes, and the per-site outputs (probabilities, posterior, decision, QUAL: 140 B/site) come back in
        pinned buffers.  Yields one dict of numpy arrays per chunk, in order."""
        from collections import deque
        copy = torch.cuda.Stream(self.device)
        main = torch.cuda.current_stream(self.device)
        inflight = deque()

        def submit(arrays, site_pos):
            with torch.cuda.stream(copy):
                dp = self.upload(arrays)
                sp = torch.as_tensor(np.ascontiguousarray(site_pos, dtype=np.int32)).pin_memory().to(self.device, non_blocking=True)
                up = torch.cuda.Event()
                up.record(copy)
            main.wait_event(up)
            out = self.run_device(dp, sp)
            host = {k: torch.empty(out[k].shape, dtype=out[k].dtype, pin_memory=True) for k in ("probs", "post", "decision", "qual")}
            for k, h in host.items():
                h.copy_(out[k], non_blocking=True)
            done = torch.cuda.Event()
            done.record(main)
            for t in dp.t.values():
                t.record_stream(main)
            sp.record_stream(main)
            inflight.append((done, host, dp, sp, out))

        def retire():
            done, host, *_ = inflight.popleft()
            done.synchronize()
            return {k: v.numpy() for k, v in host.items()}

        for arrays, site_pos in chunks:
            submit(arrays, site_pos)
            if len(inflight) > depth:
                yield retire()
        while inflight:
            yield retire()

    def run_region(self, dev_pack, lo=1, hi=2 ** 31 - 1, snv_min_af=0.05, indel_min_af=None, min_coverage=4, alt_base_num=3, min_mq=20):
        """Candidates as an internal product: STEP 1 of the reference (extract_candidates_calling's gates, with this engine's
        min_bq as run_clairs_to:1201 passes it) on a pack that holds EVERY position of a region, then the hot path (STEP 2) on the
        SNV list (K = 4) or the indel list (K = 6) of the rows in [lo, hi] - the same pack, nothing leaves HBM in between.
        Returns (site_pos int32 device tensor, outputs of run_device)."""
        from .extract_candidates_calling import extract_candidates, candidate_positions
        indel = self.K == 6
        flags, _ = extract_candidates(dev_pack, self.min_bq, min_mq, snv_min_af, (0.05 if indel_min_af is None else indel_min_af) if indel else 1.0,
                                      min_coverage, alt_base_num, indel)
        sites = candidate_positions(dev_pack, flags, 2 if indel else 1, lo, hi)
        return sites, self.run_device(dev_pack, sites)

    def run_chunk(self, arrays, site_pos, want_raw=False):
        dp = self.upload(arrays)
        sp = torch.as_tensor(np.ascontiguousarray(site_pos, dtype=np.int32)).to(self.device)
        return self.run_device(dp, sp, want_raw=want_raw)
