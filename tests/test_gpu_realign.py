"""The device form of the Illumina realigner (csrc/realign_batch.hip; SURVEY.md 8f #4b, the `realign_reads` leg of BASELINE
configs[3]): k_fast_pass (src/realign/realigner.cpp:129-229) and k_sw_ends (src/realign/ssw.c:118-529 as ssw_align drives them,
:781-830) for every window of a batch, traceback and CIGAR composition on the host - POS and CIGAR of every read byte for byte
what the reference's own realigner.cpp + SSW return (oracle/_ref/librealigner_ref.so, compiled from /root/reference by
`make -C oracle ref`; it travels to the GPU box as a built file) and what the committed golden windows hold."""
import gzip
import json
import os

import numpy as np
import pytest

import realignutil as ru

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    torch.cuda.set_device(0)
    return torch.device("cuda:0")


def test_device_realigner_on_the_golden_windows(dev):
    with gzip.open(os.path.join(HERE, "golden", "realign.json.gz"), "rb") as f:
        g = json.loads(f.read())
    rng = np.random.default_rng(g["seed"])
    ws = [ru.gen_window(rng) for _ in g["windows"]]
    st = {}
    got = ru.amd_realign_batch(ws, "device", threads=8, stats=st)
    bad = [i for i, (w, (pos, cig), want) in enumerate(zip(ws, got, g["windows"]))
           if [[p - w["ref_start"], c] for p, c in zip(pos, cig)] != want]
    assert not bad, "windows %s differ from the reference's output" % bad[:10]
    assert st["host_windows"] == 0 and st["fast_pairs"] > 10000 and st["sw_pairs"] > 2000
    print("golden windows: %d windows, %d reads, fast pass %.3f ms (%d pairs), Smith-Waterman %.3f ms (%d alignments, %.2e cells)" % (
        st["windows"], st["reads"], st["fast_pass_ms"], st["fast_pairs"], st["sw_ms"], st["sw_pairs"], st["sw_cells"]))


@pytest.mark.skipif(ru.ref_lib() is None, reason="oracle/_ref/librealigner_ref.so not built (`make -C oracle ref`, needs /root/reference)")
def test_device_realigner_equals_the_compiled_reference_on_fresh_windows(dev):
    seed = int.from_bytes(os.urandom(4), "little")
    rng = np.random.default_rng(seed)
    ws = [ru.gen_window(rng) for _ in range(400)] + [ru.gen_window(rng, n_reads=n) for n in (300, 1000)]
    got = ru.amd_realign_batch(ws, "device", threads=8)
    for i, (w, g) in enumerate(zip(ws, got)):
        assert g == ru.ref_realign(w), "window %d of np.random.default_rng(%d)" % (i, seed)


def test_device_and_host_forms_agree_and_long_windows_fall_back(dev):
    rng = np.random.default_rng(77)
    ws = [ru.gen_window(rng) for _ in range(60)]
    # a window too long for the device form (haplotypes of 2 500 bases) and one with a 600-base read: host stages inside the same call
    long_ref = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 2500))
    hap = long_ref[:1200] + "T" + long_ref[1200:]
    reads = [hap[a:a + 150] for a in (100, 1100, 1150, 2000)]
    ws.append(dict(seqs=reads, positions=[100, 1100, 1150, 2000], cigars=["150M"] * 4, reference=long_ref, haplotypes=[long_ref, hap],
                   ref_start=5000, ref_prefix=100, ref_suffix=100))
    r2 = ru.gen_window(rng, n_reads=6)
    r2["seqs"][0] = (r2["reference"] * 3)[:600]
    r2["cigars"][0] = "600M"
    ws.append(r2)
    st = {}
    got = ru.amd_realign_batch(ws, "device", threads=4, stats=st)
    assert st["host_windows"] >= 1
    assert got == ru.amd_realign_batch(ws, "host", threads=4)
    assert got[-2][1][1] != "150M"                                 # the insertion haplotype re-aligned a read of the long window
    # a batch is independent of how it is cut
    assert got[:10] == ru.amd_realign_batch(ws[:10], "device") and got[10:] == ru.amd_realign_batch(ws[10:], "device")
    assert ru.amd_realign_batch([], "device") == []


def test_device_realigner_word_mode_and_ties(dev):
    """long exact matches (the 8-bit pass overflows at score 249: 63 matching bases) and tandem repeats (equal-score cells:
    the lazy-F corrections and the first-visited rule of the fast pass decide)"""
    rng = np.random.default_rng(5)
    ws = []
    for it in range(80):
        unit = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 5))))
        core = (unit * 200)[:int(rng.integers(150, 400))]
        flank = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 120))
        ref = flank + core + flank[::-1]
        hap = flank + core[:len(core) // 2] + unit * int(rng.integers(1, 4)) + core[len(core) // 2:] + flank[::-1]
        src = [ref, hap]
        seqs, pos = [], []
        for _ in range(12):
            s = src[int(rng.integers(0, 2))]
            a = int(rng.integers(0, len(s) - 151))
            r = list(s[a:a + 150])
            for _ in range(int(rng.integers(0, 6))):
                r[int(rng.integers(0, 150))] = "ACGT"[int(rng.integers(0, 4))]
            seqs.append("".join(r))
            pos.append(1000 + a)
        ws.append(dict(seqs=seqs, positions=pos, cigars=["150M"] * 12, reference=ref, haplotypes=sorted({ref, hap}), ref_start=1000,
                       ref_prefix=100, ref_suffix=100))
    got = ru.amd_realign_batch(ws, "device", threads=4)
    assert got == ru.amd_realign_batch(ws, "host", threads=4)
    if ru.ref_lib() is not None:
        for w, g in zip(ws, got):
            assert g == ru.ref_realign(w)


from test_realign_flow import flow  # noqa: E402,F401  (the simulated paired-end run the reference's Python was recorded on)


def test_realign_variants_with_the_device_realigner_writes_the_reference_vcf(dev, flow, tmp_path):
    """`realign_variants --realigner device`: the calls of the run are worker threads whose windows meet in WindowBatcher and go to the
    device in batches - the output VCF is the one the unmodified reference wrote (tests/golden/realign_flow.json.gz), SNV and indel
    pass, and `realign_reads` SAM text through the batcher equals the one-window-at-a-time text."""
    import io
    from argparse import Namespace
    import realignsim
    from clairs_to_amd import realign_variants as rv
    from clairs_to_amd import realign_reads as rr
    g, sim, paths = flow
    for is_indel, key in ((False, "vcf"), (True, "vcf_indel")):
        out = str(tmp_path / ("device_%s.vcf" % key))
        failed = rv.realign_variants(Namespace(bam_fn=paths["bam"], ref_fn=paths["ref"], ctg_name=realignsim.CTG, pileup_vcf_fn=paths["vcf"],
                                               output_vcf_fn=out, samtools=paths["samtools"], threads=8, show_ref=False, min_mq=20, min_bq=0,
                                               enable_realignment=True, is_indel=is_indel, realigner="device"))
        assert open(out).read() == g[key]
        assert len(failed) >= 8
    pos = int(sorted(g["positions"])[0])
    a = Namespace(pos=pos, ctg_name=realignsim.CTG, bam_fn=paths["bam"], ref_fn=paths["ref"], samtools=paths["samtools"], min_mq=20,
                  min_coverage=2.0, realign_flanking_window=100, max_distance=50)
    want = io.StringIO()
    rr.reads_realignment(a, out=want)
    with rr.WindowBatcher("device", threads=2) as b:
        a.realign_fn = b
        got = io.StringIO()
        rr.reads_realignment(a, out=got)
    assert got.getvalue() == want.getvalue() and b.windows >= 1


def test_device_sw_passes_on_adversarial_pairs(dev):
    """cto_sw_ends_batch on the device - the kernel whose lazy-F step is the closed form (csrc/realign_batch.hip) - on pairs built to
    stress exactly that step (long matches around long gaps in both orientations, tandem repeats, tiny alphabets, queries shorter than
    the lane count), operands up to 2 000 bases: every end point equal to the scalar model's (oracle/ssw_model.cpp, the loops as the
    reference has them) on 3 000 pairs and to the SSE2 host form on 20 000 more, fresh pairs every run."""
    from clairs_to_amd.realign_reads import sw_ends_batch
    seed = int.from_bytes(os.urandom(4), "little")
    rng = np.random.default_rng(seed)
    small = ru.adversarial_pairs(rng, 3000, max_len=500)
    small += [(np.zeros(0, dtype=np.int8), small[0][1]), (small[0][0], np.zeros(0, dtype=np.int8))]
    got = sw_ends_batch(small, "device")
    for i, ((ref, q), o) in enumerate(zip(small, got)):
        assert o.tolist() == ru.model_ends(ref, q), "pair %d of np.random.default_rng(%d)" % (i, seed)
    big = ru.adversarial_pairs(rng, 20000, max_len=700) + ru.adversarial_pairs(rng, 600, max_len=2000)
    d, h = sw_ends_batch(big, "device"), sw_ends_batch(big, "host", threads=16)
    bad = np.nonzero((d != h).any(axis=1))[0]
    assert bad.size == 0, "pair %d of np.random.default_rng(%d): device %s host %s" % (bad[0], seed, d[bad[0]], h[bad[0]])
    assert int((d[:, 5] == 8).sum()) > 2000 and int((d[:, 0] > 1000).sum()) > 500


def test_device_alignments_on_adversarial_pairs(dev):
    """cto_ssw_align_batch on the device (k_sw for the end points, k_banded for the traceback) against the host form - score, reference
    start and CIGAR - on pairs built to stress the band: long matches around gaps of up to 80 bases in both orientations (the band
    starts at the length difference and doubles), several gaps (the walk changes state often), tandem repeats (ties in every cell),
    unrelated pairs (short local alignments), operands up to 2 000 bases; fresh pairs every run."""
    from clairs_to_amd.realign_reads import ssw_align_batch
    seed = int.from_bytes(os.urandom(4), "little")
    rng = np.random.default_rng(seed)
    pairs = ru.adversarial_pairs(rng, 12000, max_len=700) + ru.adversarial_pairs(rng, 400, max_len=2000)
    pairs += [(np.zeros(5, dtype=np.int8), np.ones(7, dtype=np.int8)), (np.zeros(0, dtype=np.int8), np.ones(7, dtype=np.int8))]
    ds, db, dc = ssw_align_batch(pairs, "device", threads=16)
    hs, hb, hc = ssw_align_batch(pairs, "host", threads=16)
    for i in range(len(pairs)):
        assert (int(ds[i]), int(db[i]), dc[i]) == (int(hs[i]), int(hb[i]), hc[i]), "pair %d of np.random.default_rng(%d)" % (i, seed)
    gapped = sum(1 for c in dc if "D" in c or "I" in c)
    assert gapped > 3000 and sum(1 for c in dc if c == "") >= 2


def test_device_sw_register_form_at_its_boundaries_and_known_overflows(dev):
    """The register form of the passes (csrc/realign_batch.hip: row_pass_regs - NG groups of eight stripe positions per lane, the
    positions behind a stripe's end computed and inert) at every boundary it has: query lengths around 8 / 16 lanes x 4, 8, 16, 32, 64,
    96 and 128 positions (one below, at, one above: the last group exactly full, one position in the next group, the class changes),
    long and short references; and the 8-bit pass's early exit (sure_overflow16): operands that share their first or their last 64
    bases and nothing else, 62 / 63 shared bases (no exit: the pass itself decides), an N among the 64 - against the SSE2 host form."""
    from clairs_to_amd.realign_reads import sw_ends_batch
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    pairs = []
    for lanes in (8, 16):
        for seg in (4, 8, 16, 32, 64, 96, 128):
            for dq in (-1, 0, 1, 7):
                Q = lanes * seg + dq
                if Q > 2040:
                    continue
                for R in (Q // 2 + 3, Q + 40):
                    ref = rng.integers(0, 4, R).astype(np.int8)
                    q = np.concatenate([ref[: min(R, Q) // 2], rng.integers(0, 4, Q - min(R, Q) // 2).astype(np.int8)])       # half a match, half noise
                    pairs.append((ref, q))
                    q2 = np.tile(np.array([0, 1], dtype=np.int8), Q // 2 + 1)[:Q].copy()                                     # ties in every cell
                    pairs.append((np.tile(np.array([0, 1], dtype=np.int8), R // 2 + 1)[:R].copy(), q2))
    for n_same in (62, 63, 64, 65, 200):
        for at_end in (False, True):
            for with_n in (False, True):
                a = rng.integers(0, 4, n_same).astype(np.int8)
                if with_n:
                    a[n_same // 2] = 4
                x, y = rng.integers(0, 4, int(rng.integers(80, 400))).astype(np.int8), rng.integers(0, 4, int(rng.integers(80, 400))).astype(np.int8)
                pairs.append((np.concatenate([x, a]), np.concatenate([y, a])) if at_end else (np.concatenate([a, x]), np.concatenate([a, y])))
    d, h = sw_ends_batch(pairs, "device"), sw_ends_batch(pairs, "host", threads=16)
    bad = np.nonzero((d != h).any(axis=1))[0]
    assert bad.size == 0, "pair %d: device %s host %s" % (bad[0], d[bad[0]], h[bad[0]])
    assert int((d[:, 5] == 8).sum()) > 40 and int((d[:, 5] == 16).sum()) > 40          # both the 16-bit and the 8-bit results are in the set


def test_device_sw_on_thousands_of_longest_queries(dev):
    """a class of more than 4 096 alignments whose queries are as long as the device form takes (2 048 bases): eight 8-lane rows of
    them do not fit the LDS of a workgroup - the launch must share it among fewer rows, not refuse the call"""
    from clairs_to_amd.realign_reads import sw_ends_batch
    rng = np.random.default_rng(12)
    ref = rng.integers(0, 4, 2048).astype(np.int8)
    pairs = []
    for i in range(4300):
        q = ref.copy()
        k = int(rng.integers(100, 1900))
        q[k] = (q[k] + 1) % 4
        if i % 3 == 0:
            q = np.concatenate([q[:k], q[k + int(rng.integers(1, 40)):]])
        pairs.append((ref, q))
    d = sw_ends_batch(pairs, "device")
    h = sw_ends_batch(pairs[:600], "host", threads=16)
    assert (d[:600] == h).all() and int((d[:, 5] == 8).sum()) == len(pairs)


def test_concurrent_device_calls_give_what_serial_calls_give(dev):
    """two host threads inside cto_realign_windows at once (what two dispatchers of one process would do): the kept direction
    scratch serves one of them, the other allocates its own; launch attributes are set once, not per call"""
    import threading
    rng = np.random.default_rng(9)
    batches = [[ru.gen_window(rng) for _ in range(120)] for _ in range(4)]
    want = [ru.amd_realign_batch(b, "device", threads=4) for b in batches]
    got = [None] * len(batches)

    def work(i):
        import torch
        torch.cuda.set_device(0)
        for _ in range(3):
            got[i] = ru.amd_realign_batch(batches[i], "device", threads=4)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(batches))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert got == want
