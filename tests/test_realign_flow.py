"""The Python around the Illumina realigner (SURVEY.md 8f #4b): clairs_to_amd.realign_reads / realign_variants against what the
REFERENCE's src/realign_reads.py and src/realign_variants.py produced on the simulated short-read data of
tests/golden/realignsim.py (tests/golden/realign_flow.json.gz, written by tests/golden/gen_realign.py flow: the reference run
unmodified, its realigner compiled from its own sources, `samtools` = the shim of realignsim.py, the consensus strings from the
build's de Bruijn graph recorded as inputs - the reference's own needs Boost, PARITY UNPINNED for that one piece)."""
import gzip
import hashlib
import io
import json
import os
import sys
from argparse import Namespace

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import realignsim  # noqa: E402


@pytest.fixture(scope="module")
def flow(tmp_path_factory):
    with gzip.open(os.path.join(HERE, "golden", "realign_flow.json.gz"), "rb") as f:
        g = json.loads(f.read())
    d = str(tmp_path_factory.mktemp("realign_flow"))
    sim = realignsim.simulate(g["seed"])
    paths = realignsim.write_inputs(sim, d)
    h = hashlib.sha256()
    for k in ("ref", "vcf"):
        h.update(open(paths[k], "rb").read())
    h.update(open(paths["bam"] + ".sam", "rb").read())
    assert h.hexdigest() == g["inputs_sha256"], "the simulator changed: regenerate tests/golden/realign_flow.json.gz"
    os.environ["REALIGNSIM_TESTS"] = HERE
    return g, sim, paths


def _args(paths, pos):
    return Namespace(pos=pos, ctg_name=realignsim.CTG, bam_fn=paths["bam"], ref_fn=paths["ref"], samtools=paths["samtools"], min_mq=20,
                     min_coverage=2.0, realign_flanking_window=100, max_distance=50)


def _orig(sim):
    return {(r["name"] + "_" + str(int(bool(r["flag"] & 16))), r["flag"]): (r["pos"], "".join("%d%s" % (n, o) for o, n in r["cigar"]))
            for r in sim["reads"]}


def test_realign_reads_writes_the_reference_sam_text(flow):
    from clairs_to_amd import realign_reads as rr
    g, sim, paths = flow
    orig = _orig(sim)
    moved_total = 0
    for pos, want in g["positions"].items():
        out = io.StringIO()
        rr.reads_realignment(_args(paths, int(pos)), out=out)
        text = out.getvalue()
        rows = [r for r in text.split("\n") if r and r[0] != "@"]
        moved = []
        for r in rows:
            c = r.split("\t")
            if orig[(c[0], int(c[1]))] != (int(c[3]) - 1, c[5]):
                moved.append([c[0], int(c[3]), c[5]])
        assert moved == want["moved"], pos
        assert len(rows) == want["n_rows"]
        assert hashlib.sha256(text.encode()).hexdigest() == want["sha256"], pos
        moved_total += len(moved)
    assert len(g["positions"]) >= 20 and moved_total >= 200


def test_realign_reads_across_its_chunk_boundary(flow):
    """--realign_flanking_window 3000: 8 200 bases of reads cross the entry point's 5 000-base chunk, so rows are written and evidence
    is dropped at the boundary (src/realign_reads.py:283-286, 618-634) - a path the filter's own window of 100 never takes"""
    from clairs_to_amd import realign_reads as rr
    g, sim, paths = flow
    for pos, want in g["wide_window"].items():
        a = _args(paths, int(pos))
        a.realign_flanking_window = 3000
        out = io.StringIO()
        rr.reads_realignment(a, out=out)
        text = out.getvalue()
        assert len([r for r in text.split("\n") if r and r[0] != "@"]) == want["n_rows"]
        assert hashlib.sha256(text.encode()).hexdigest() == want["sha256"], pos
    assert len(g["wide_window"]) == 3 and sum(w["n_moved"] for w in g["wide_window"].values()) > 5


def test_realign_reads_with_the_recorded_consensus_as_input(flow):
    """the part that is pinned without the build's de Bruijn graph: the recorded haplotypes go in, the reference's SAM text comes
    out; and the graph, asked again, returns what was recorded"""
    from clairs_to_amd import realign_reads as rr
    g, sim, paths = flow
    log = {}
    for pos, ref_w, haps in g["consensus"]:
        log.setdefault(pos, []).append((ref_w, haps))
    for pos, want in g["positions"].items():
        calls = list(log.get(int(pos), []))

        def replay(centre, reads, lowbq):
            ref_w, haps = calls.pop(0)
            assert ref_w == centre
            assert rr.dbg_consensus(centre, reads, lowbq) == haps
            return haps
        (rd_lo, rd_hi), (ref_lo, ref_hi) = rr.region_of(int(pos), 100)
        ref = rr.faidx(paths["samtools"], paths["ref"], "%s:%d-%d" % (realignsim.CTG, ref_lo, ref_hi))
        rows = [r + "\n" for r in realignsim.sam_text(sim).split("\n") if r]
        keep = []
        for r in rows:
            if r[0] == "@":
                keep.append(r)
                continue
            rd = realignsim.parse_sam(r)[0]
            end = rd["pos"] + sum(n for o, n in rd["cigar"] if o in "MDN=X")
            if rd["mapq"] >= 20 and rd["pos"] < rd_hi and end > rd_lo - 1:
                keep.append(r)
        out = io.StringIO()
        rr.realign_region(keep, realignsim.CTG, ref, ref_lo - 1, int(pos), out, 2.0, 50, consensus_fn=replay)
        assert not calls
        assert hashlib.sha256(out.getvalue().encode()).hexdigest() == want["sha256"], pos


def test_realign_variants_writes_the_reference_vcf(flow, tmp_path):
    from clairs_to_amd import realign_variants as rv
    g, sim, paths = flow
    out = str(tmp_path / "out" / "realigned.vcf")
    failed = rv.realign_variants(Namespace(bam_fn=paths["bam"], ref_fn=paths["ref"], ctg_name=realignsim.CTG, pileup_vcf_fn=paths["vcf"],
                                           output_vcf_fn=out, samtools=paths["samtools"], threads=8, show_ref=False, min_mq=20, min_bq=0,
                                           enable_realignment=True, is_indel=False))
    assert open(out).read() == g["vcf"]
    assert len(failed) == g["vcf"].count("LowQual;Realignment") >= 8
    # the same calls in worker processes (--pool process: what the reference's ProcessPoolExecutor does): the same file
    out_p = str(tmp_path / "out" / "realigned_processes.vcf")
    rv.realign_variants(Namespace(bam_fn=paths["bam"], ref_fn=paths["ref"], ctg_name=realignsim.CTG, pileup_vcf_fn=paths["vcf"], output_vcf_fn=out_p,
                                  samtools=paths["samtools"], threads=4, pool="process", show_ref=False, min_mq=20, min_bq=0,
                                  enable_realignment=True, is_indel=False))
    assert open(out_p).read() == g["vcf"]
    # the indel pass: indel records stay in; a deletion's ALT is its bare anchor base, which the reference counts like an SNV allele
    out_i = str(tmp_path / "out" / "realigned_indel.vcf")
    rv.realign_variants(Namespace(bam_fn=paths["bam"], ref_fn=paths["ref"], ctg_name=realignsim.CTG, pileup_vcf_fn=paths["vcf"], output_vcf_fn=out_i,
                                  samtools=paths["samtools"], threads=8, show_ref=False, min_mq=20, min_bq=0, enable_realignment=True, is_indel=True))
    assert open(out_i).read() == g["vcf_indel"] and g["vcf_indel"].count("\n") > g["vcf"].count("\n")
    # the switch (--enable_realignment False): a link to the input, as the reference leaves it (src/realign_variants.py:134-136)
    off = str(tmp_path / "off.vcf")
    rv.realign_variants(Namespace(enable_realignment=False, pileup_vcf_fn=paths["vcf"], output_vcf_fn=off))
    assert os.path.islink(off) and open(off).read() == open(paths["vcf"]).read()


def test_column_alleles_and_decision_rule():
    """get_base_list / the demotion rule (src/realign_variants.py:31-56, :112-123), hand-derived"""
    from clairs_to_amd.realign_variants import column_alleles, decide
    assert column_alleles("A$a^]C+2ACg-1nT*#>N") == ["A", "A", "C+AC", "G-N", "T", "*", "#", "N"]      # the length digits are dropped
    assert decide("AAAATTTT", "AAAAAATT", "T") == (False, (4, 8, 2, 8))         # fewer reads and a smaller fraction: demoted
    assert decide("AAAATTTT", "AATTTT", "T")[0] is True                          # same support, smaller depth
    assert decide("AAAATTTT", "TT", "T")[0] is True                              # fewer reads but a larger fraction
    assert decide("AAAA", "AAAA", "T")[0] is True


def test_realign_variants_without_samtools(flow, tmp_path):
    """--bam_reader native: the same decisions from a real BAM + BAI (tests/bamutil.write_bam) with no samtools anywhere - the
    built-in reader serves the raw column and the region's rows (cto_pack_from_bam, cto_bam_view), the realigned column is piled up
    in-process.  The VCF equals the one the reference wrote through the samtools shim (whose pileup is the naive restatement the
    built-in reader is held to): PARITY UNPINNED against samtools itself."""
    from bamutil import write_bam
    from clairs_to_amd import realign_variants as rv
    g, sim, paths = flow
    bam = str(tmp_path / "sim.bam")
    reads = [dict(r, cigar=[(o, n) for o, n in r["cigar"]]) for r in sim["reads"]]
    write_bam(bam, [(realignsim.CTG, len(sim["ref"]))], reads, block_payload=20000)
    out = str(tmp_path / "native.vcf")
    failed = rv.realign_variants(Namespace(bam_fn=bam, ref_fn=paths["ref"], ctg_name=realignsim.CTG, pileup_vcf_fn=paths["vcf"], output_vcf_fn=out,
                                           samtools="/nonexistent/samtools", threads=8, show_ref=False, min_mq=20, min_bq=0,
                                           enable_realignment=True, is_indel=False, bam_reader="native"))
    assert open(out).read() == g["vcf"]
    assert len(failed) >= 8


def test_evidence_scan_in_c_equals_the_python_loop(flow, monkeypatch):
    """RegionRealigner.feed counts a read's evidence in C (cto_realign_read_evidence) and keeps the Python loop for whatever the C
    side declines: the same SAM text with the C scan switched off, and the same counters on reads with every CIGAR operation."""
    import random
    from clairs_to_amd import realign_reads as rr
    g, sim, paths = flow
    pos = int(sorted(g["positions"])[0])
    want = io.StringIO()
    rr.reads_realignment(_args(paths, pos), out=want)

    real = rr.lib

    class NoScan(object):
        def __getattr__(self, name):
            if name == "cto_realign_read_evidence":
                return lambda *a: -1
            return getattr(real, name)
    monkeypatch.setattr(rr, "lib", NoScan())
    got = io.StringIO()
    rr.reads_realignment(_args(paths, pos), out=got)
    monkeypatch.setattr(rr, "lib", real)
    assert got.getvalue() == want.getvalue() and want.getvalue().count("\n") > 100

    rnd = random.Random(4)
    ref = "".join(rnd.choice("ACGTN") for _ in range(3000))
    for trial in range(300):
        ops, qlen = [], 0
        for _ in range(rnd.randint(1, 7)):
            op, n = rnd.choice("MMM=XIDSNH"), rnd.randint(1, 30)
            ops.append("%d%s" % (n, op))
            qlen += n if op in "M=XIS" else 0
        cigar = "".join(ops)
        seq = "".join(rnd.choice("ACGT") for _ in range(max(qlen, 1)))
        bq = "".join(chr(33 + rnd.choice((5, 19, 20, 21, 35))) for _ in seq)
        start = rnd.randint(1100, 1500)
        row = "r%d\t%d\t%s\t%d\t60\t%s\t=\t1\t0\t%s\t%s\n" % (trial, rnd.choice((0, 16)), realignsim.CTG, start + 1, cigar, seq, bq)
        a = rr.RegionRealigner(realignsim.CTG, ref, 1000, 1300)
        a.feed(row, io.StringIO())
        monkeypatch.setattr(rr, "lib", NoScan())
        b = rr.RegionRealigner(realignsim.CTG, ref, 1000, 1300)
        b.feed(row, io.StringIO())
        monkeypatch.setattr(rr, "lib", real)
        assert a.evidence == b.evidence, cigar
