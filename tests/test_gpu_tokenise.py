"""mpileup text -> pack on the device (csrc/tokenise.hip: cto_tokenise_device; SURVEY.md 8a F2 + F9 / F10, the tokeniser of
src/create_tensor_pileup_calling.py:120-144 and the row handling of :465-532) against cto_pack_from_mpileup, the host tokeniser that is
itself pinned to the reference's decode_pileup_bases (tests/test_oracle_golden.py, tests/test_gpu_parity.py): every array and every
alt_info key string equal; rows the single pass does not take come back as a fallback, never as a different pack."""
import numpy as np
import pytest

from conftest import load_json_gz

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    torch.cuda.set_device(0)
    return torch.device("cuda:0")


def _same_pack(text, ref, ref_start, max_indel=60):
    from clairs_to_amd.pack import ColumnPack, DeviceTokeniser
    host = ColumnPack.from_mpileup(text, ref, ref_start, max_indel)
    tok = DeviceTokeniser()
    got = tok(text, ref, ref_start, max_indel)
    assert got is not None, "the device tokeniser declined a text the single pass takes"
    view, lite = got
    a, b = host.numpy(), DeviceTokeniser.download(view)
    assert (view.n_cols, view.n_entries, view.n_keys) == (host.n_cols, host.n_entries, host.n_keys)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    ln = lite.numpy()
    for k in ("col_pos", "col_ref", "key_off", "key_meta", "key_group"):
        np.testing.assert_array_equal(a[k], ln[k], err_msg="lite " + k)
    for k in range(host.n_keys):
        assert lite.key_string(k) == host.key_string(k), k
    return host


def test_golden_region_text(dev, golden_region):
    """the mpileup text the reference's own create_tensor consumed for tests/golden/region.json.gz (both --min_bq passes)"""
    g = golden_region
    for key in ("mpileup_aff", "mpileup_neg", "mpileup"):
        if key in g:
            h = _same_pack(g[key], g["ref"], g["ref_start"])
            assert h.n_cols > 100


def test_synthetic_chunks_every_platform(dev):
    """a 4096-site ONT chunk (22 MB of text, 140 000 rows) and HiFi / Illumina presets: indels up to 80 bases (overlong keys),
    '*' / '#', N bases, '^x' and '$', MQ / BQ over the whole printable range"""
    import oracle
    from clairs_to_amd.synth import SynthChunk
    for platform, n in (("ont", 4096), ("hifi", 512), ("ilmn", 512)):
        ch = SynthChunk.for_platform(platform, n) if platform != "ont" else SynthChunk(n, seed=3)
        text = oracle.synth_mpileup_text(ch, 0)
        ref, lo = ch.ref_window()
        h = _same_pack(text, ref, lo)
        assert h.n_keys > 100 and h.n_entries > 10000


def test_columns_fixture_rows(dev):
    """the hand-made and random columns of tests/golden/columns.json.gz (SURVEY 8a F2-F6 edge cases: 59/60-base deletions, 60/61-base
    insertions, '*+', '#+', N, '^x', '$', doubled indel annotations) as rows of one text"""
    cases = load_json_gz("columns.json.gz")
    ref = "ACGT" * 4000
    rows, pos = [], 100
    for c in cases:
        bases, bq, mq = c["bases"], c["bq"], c["mq"]
        rows.append("chr1\t%d\t%s\t%d\t%s\t%s\t%s\n" % (pos, "N", len(bq), bases, bq, mq))
        pos += 3
    from clairs_to_amd.pack import ColumnPack, DeviceTokeniser
    text = "".join(rows)
    host = ColumnPack.from_mpileup(text, ref, 1)
    got = DeviceTokeniser()(text, ref, 1)
    if got is None:
        # some fixture columns are deliberately malformed (short quality strings ...): those rows are the host's; the rest must agree
        good = []
        for r in rows:
            if DeviceTokeniser()(r, ref, 1) is not None:
                good.append(r)
        assert len(good) > 0.8 * len(rows)
        _same_pack("".join(good), ref, 1)
    else:
        _same_pack(text, ref, 1)
    assert host.n_cols > 100


def test_what_the_single_pass_declines_falls_back(dev):
    from clairs_to_amd.pack import DeviceTokeniser
    tok = DeviceTokeniser()
    ref = "ACGT" * 100
    ok = "chr1\t10\tN\t3\tAc+2gt*\tIII\t]]]\n"
    assert tok(ok, ref, 1) is not None
    for bad in (ok[:-1],                                              # no final newline
                ok.replace("\n", "\r\n"),                            # CR LF
                "chr1\t10\tN\t3\tAc*\tII\t]]]\n",                   # short quality string
                "chr1\t10\tN\t3\tAc*\tIII\t]]\n",                    # short mapping-quality string
                "chr1\t10\tN\t3\tAc*\tIII\n",                        # six fields
                "chr1\t10\tN\t3\tAc*\tI\x07I\t]]]\n",               # a control character in the qualities
                "chr1\t10\tN\t3\tAc+5gt\tII\t]]\n",                 # an indel running into the field's end
                ok + "\n",                                           # an empty row
                "chr1\t10\tN\t1\tA\tI\t]\nchr1\t9\tN\t1\tA\tI\t]\n",   # rows out of order
                "chr1\t5000\tN\t1\tA\tI\t]\n",                      # outside the reference slice
                "chr1\t10\tN\t300\t" + "A+1c" * 300 + "\t" + "I" * 300 + "\t" + "]" * 300 + "\n",     # 300 indel carriers in a row (the limit is 256)
                ""):
        assert tok(bad, ref, 1) is None, repr(bad[:60])
    # ... and the context is still good afterwards
    assert tok(ok, ref, 1) is not None


def test_rows_with_many_indel_carriers(dev):
    """round 6: a row's indel tokens are a chain of records in HBM, not a lane's private array - 40 carriers in a row (the old form declined
    more than 32), 200 carriers with a dozen distinct keys in both cases and on both strands, doubled annotations, over-long ones"""
    ref = "ACGT" * 200
    rng = np.random.default_rng(11)
    rows = []
    for pos, n in ((10, 40), (11, 200), (12, 3), (13, 256)):
        toks = []
        for i in range(n):
            b = "ACGTacgt*#"[int(rng.integers(0, 10))]
            k = int(rng.integers(0, 6))
            t = b + ("", "+1c", "-2at", "+3GGT", "+1c-1g", "+61" + "a" * 61)[k] if b not in "*#" or k in (0, 1, 3) else b
            toks.append(t)
        rows.append("chr1\t%d\tN\t%d\t%s\t%s\t%s\n" % (pos, n, "".join(toks), "I" * n, "]" * n))
    _same_pack("".join(rows), ref, 1)


def test_deep_columns_take_the_unstaged_path(dev):
    """a wavefront's 64 rows are staged in LDS when they fit 16 KB; deeper columns (the generator caps a column at 200 reads: ~600 bytes per row, 38 KB per wavefront) are walked from
    HBM with the register byte stream and write their codes straight to the codes buffer - same pack; mixed with shallow rows in one text"""
    import oracle
    from clairs_to_amd.synth import SynthChunk
    rows = []
    ref_parts = {}
    for k, (depth, n) in enumerate(((400.0, 96), (30.0, 200), (2000.0, 24))):
        ch = SynthChunk(n, seed=40 + k, depth_mean=depth, start=1000 + k * 200000, spacing=3, p_ins=0.01, p_del=0.01)
        text = oracle.synth_mpileup_text(ch, 0)
        rows.append(text if isinstance(text, str) else text.decode())
        r, lo = ch.ref_window()
        ref_parts[lo] = r
    lo0 = min(ref_parts)
    hi0 = max(lo + len(r) for lo, r in ref_parts.items())
    ref = bytearray(b"A" * (hi0 - lo0))
    for lo, r in ref_parts.items():
        ref[lo - lo0:lo - lo0 + len(r)] = r.encode()
    h = _same_pack("".join(rows), ref.decode(), lo0)
    assert h.n_entries > 90000 and h.n_keys > 200


def test_tensors_from_a_device_tokenised_pack(dev, golden_region):
    """featurisation fed from the device-born pack = featurisation fed from the uploaded host pack (bit for bit)"""
    import ctypes as C
    import torch
    from clairs_to_amd._lib import lib, check, current_stream_ptr
    from clairs_to_amd.featurize import featurize
    from clairs_to_amd.pack import ColumnPack, DeviceTokeniser
    g = golden_region
    key = "mpileup_neg"
    host = ColumnPack.from_mpileup(g[key], g["ref"], g["ref_start"])
    sites = torch.tensor(sorted(int(x) for x in g["sites"]), dtype=torch.int32, device=dev)
    want = featurize(host.to_device(dev), sites, 20, 50, want_raw=True)
    view, lite = DeviceTokeniser()(g[key], g["ref"], g["ref_start"])

    class Shim(object):           # what featurize() reads of a DevicePack
        pass
    shim = Shim()
    shim.view, shim.n_cols, shim.n_entries, shim.n_keys, shim.host, shim.device = view, int(view.n_cols), int(view.n_entries), int(view.n_keys), lite, dev
    got = featurize(shim, sites, 20, 50, want_raw=True)
    torch.cuda.synchronize()
    assert torch.equal(got.raw_aff, want.raw_aff) and torch.equal(got.raw_neg, want.raw_neg) and torch.equal(got.x_aff, want.x_aff)
    assert torch.equal(got.site_info, want.site_info)
