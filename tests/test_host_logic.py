"""CPU-only tests of the host side: C-ABI loads and exports every declared symbol, the mpileup tokeniser
round-trips the synthetic packs, and the VCF row assembly reproduces the reference's call_variants output when
fed the reference's own probability rows (posterior from the CPU oracle here; from the GPU in test_gpu_parity)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_json_gz


def test_cabi_exports_every_declared_symbol():
    import clairs_to_amd._lib as L
    header = open(os.path.join(ROOT, "include", "clairsto_amd.h")).read()
    declared = set(re.findall(r"\b(cto_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    for name in declared:
        assert hasattr(L.lib, name)
    assert L.lib.cto_version() >= 100


def test_mpileup_tokeniser_round_trips_synthetic_pack():
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd.synth import SynthChunk, mpileup_text
    chunk = SynthChunk(40, seed=9, p_ins=0.03, p_del=0.04, spacing=30)
    ref, lo = chunk.ref_window()
    pack = ColumnPack.from_mpileup(mpileup_text(chunk), ref, lo)
    got, want = pack.numpy(), chunk.arrays()
    for k in ("col_pos", "col_ref", "col_off", "key_off", "key_meta", "key_group"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    # MQ is printed clamped to 93 ('~') by the text writer; compare entries with MQ saturated the same way
    e = want["entries"].astype(np.uint64)
    mq = np.minimum((e >> 13) & 255, 93)
    e = (e & ~np.uint64(255 << 13)) | (mq << 13)
    np.testing.assert_array_equal(got["entries"].astype(np.uint64), e)
    assert pack.n_keys > 0
    s = pack.key_string(0)
    assert s[0] in "ID" and len(s) >= 2


def test_tokeniser_rejects_malformed_rows():
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd._lib import CtoError
    with pytest.raises(CtoError):
        ColumnPack.from_mpileup("chr1\t5\tN\t1\tA\tI\n", "ACGTACGT", 1)            # no MQ column
    with pytest.raises(CtoError):
        ColumnPack.from_mpileup("chr1\t50\tN\t1\tA\tI\t]\n", "ACGTACGT", 1)        # outside the reference
    with pytest.raises(CtoError):
        ColumnPack.from_mpileup("chr1\t5\tN\t1\tA\tI\t]\nchr1\t4\tN\t1\tA\tI\t]\n", "ACGTACGT", 1)   # unsorted
    p = ColumnPack.from_mpileup("", "ACGT", 1)
    assert p.n_cols == 0 and p.n_entries == 0


def test_tokeniser_paths_agree():
    """cto_pack_from_mpileup parses a row samtools-style in one forward pass and hands everything unusual to the general
    field-splitting parser (csrc/pack.cpp fast_row / parse_rows).  The same pileup written so that every row takes the general path
    (an eighth field, or CR LF line ends) must give the same pack as the plain form, which takes the fast path; and a text cut
    over several tokeniser threads (> 4 MB) the same pack as one thread."""
    import os
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd.synth import SynthChunk, mpileup_text
    chunk = SynthChunk(300, seed=21, p_ins=0.05, p_del=0.05, spacing=40)
    ref, lo = chunk.ref_window()
    text = mpileup_text(chunk)
    text = text if isinstance(text, str) else text.decode()
    rows = [r for r in text.split("\n") if r]
    forms = {"plain": "\n".join(rows) + "\n",
             "eight_fields": "\n".join(r + "\tx" for r in rows) + "\n",
             "crlf": "\r\n".join(rows) + "\r\n",
             "no_final_newline": "\n".join(rows)}

    def arrays(t):
        p = ColumnPack.from_mpileup(t, ref, lo)
        a = {k: v.copy() for k, v in p.numpy().items()}
        a["keys"] = np.array([p.key_string(k) for k in range(p.n_keys)])
        return a
    want = arrays(forms["plain"])
    assert want["entries"].size > 5000 and want["keys"].size > 50
    for name, t in forms.items():
        got = arrays(t)
        for k in want:
            np.testing.assert_array_equal(got[k], want[k], err_msg="%s: %s" % (name, k))
    # a row whose quality string is one character short: zip() truncation of the reference (general path), not an error
    f = rows[0].split("\t")
    n0 = len(f[5])
    short = "\t".join(f[:5] + [f[5][:-1], f[6]])
    p = ColumnPack.from_mpileup(short + "\n", ref, lo)
    assert p.n_entries == n0 - 1
    # several tokeniser threads
    big = SynthChunk(4096, seed=22, spacing=300)
    bref, blo = big.ref_window()
    btext = mpileup_text(big)
    assert len(btext) > (1 << 22)
    old = os.environ.get("CTO_PACK_THREADS")
    try:
        out = []
        for nt in ("1", "7"):
            os.environ["CTO_PACK_THREADS"] = nt
            p = ColumnPack.from_mpileup(btext, bref, blo)
            out.append({k: v.copy() for k, v in p.numpy().items()})
    finally:
        if old is None:
            os.environ.pop("CTO_PACK_THREADS", None)
        else:
            os.environ["CTO_PACK_THREADS"] = old
    for k in out[0]:
        np.testing.assert_array_equal(out[0][k], out[1][k], err_msg="threads: " + k)
    # the per-thread setting (what the chunk pipelines use instead of the environment variable) wins over it, 0 hands control back
    from clairs_to_amd._lib import lib
    try:
        lib.cto_set_pack_threads(5)
        p5 = ColumnPack.from_mpileup(btext, bref, blo)
        a5 = {k: v.copy() for k, v in p5.numpy().items()}
    finally:
        lib.cto_set_pack_threads(0)
    for k in out[0]:
        np.testing.assert_array_equal(out[0][k], a5[k], err_msg="cto_set_pack_threads: " + k)


def _rows(text):
    return [r.split("\t") for r in text.strip().split("\n") if r]


@pytest.mark.parametrize("mode", ["snv", "indel"])
@pytest.mark.parametrize("show_ref", [False, True])
def test_vcf_rows_match_reference(oracle_lib, mode, show_ref):
    from clairs_to_amd.call_variants import load_likelihood, vcf_row
    calls = load_json_gz("calls_%s.json.gz" % mode)
    K = calls["n_out"]
    rows = _rows(calls["predict_rows"])
    lik, edges = load_likelihood(np.loadtxt(calls["likelihood_table"].split("\n")), K)
    p1 = np.array([[float(f.split()[1]) for f in r[6:6 + 2 * K]] for r in rows], dtype=np.float64)
    post, dec, qual = oracle_lib.posterior_from_probs(p1, lik, edges)
    assert not dec[:, 1].any()
    out = []
    for i, r in enumerate(rows):
        row = vcf_row(r[0], r[1], r[2], r[3], eval(r[4]), eval(r[5]), int(dec[i, 0]), float(qual[i]), K, show_ref=show_ref)
        if row is not None:
            out.append(row)
    want = calls["vcf"]["show_ref" if show_ref else "default"]
    assert len(want) > 0
    assert out == want


def test_sort_and_postprocess_vcf_match_reference(tmp_path):
    """chunk VCF merge + QUAL/AF gates byte-identical to the reference's sort_vcf / postprocess_vcf outputs
    (tests/golden/post.json.gz: several contigs incl. a chr1/chr11 name clash, `H` records, NonSomatic / LowQual, 6 option sets)."""
    from clairs_to_amd.postprocess_vcf import sort_vcf_main, postprocess_vcf_main
    g = load_json_gz("post.json.gz")
    d = tmp_path / "vcf_output"
    d.mkdir()
    for fn, text in g["chunks"].items():
        (d / fn).write_text(text)
    (tmp_path / "CONTIGS").write_text("".join(c + "\n" for c in g["contigs_order"]))
    (tmp_path / "ref.fa.fai").write_text(g["fai"])
    (tmp_path / "CMD").write_text(g["cmd"])
    merged = tmp_path / "merged.vcf"
    base = ["--input_dir", str(d), "--ref_fn", str(tmp_path / "ref.fa"), "--contigs_fn", str(tmp_path / "CONTIGS")]
    n = sort_vcf_main(base + ["--vcf_fn_prefix", "p_", "--output_fn", str(merged)])
    assert merged.read_text() == g["merged"] and n == sum(1 for r in g["merged"].split("\n") if r and r[0] != "#")
    empty = tmp_path / "empty.vcf"
    assert sort_vcf_main(base + ["--vcf_fn_prefix", "nothing_", "--output_fn", str(empty), "--sample_name", "S2"]) == 0
    # no records: a bare header (own meta lines, so compare the parts both sides must agree on) without a trailing newline
    from clairs_to_amd.call_variants import VCF_HEADER
    tail = [r for r in g["empty"].split("\n") if r.startswith("##contig") or r.startswith("#CHROM")]
    assert empty.read_text() == VCF_HEADER + "\n".join(tail)
    assert [r for r in VCF_HEADER.split("\n") if "ID=TU," in r] == [r for r in g["empty"].split("\n") if "FORMAT=<ID=TU," in r]
    for i, case in enumerate(g["cases"]):
        o = case["opts"]
        out = tmp_path / ("final_%d.vcf" % i)
        argv = ["--pileup_vcf_fn", str(merged), "--output_fn", str(out), "--platform", o["platform"]]
        for k in ("qual", "af", "qual_cutoff_phaseable_region", "qual_cutoff_unphaseable_region", "max_qual_filter_pileup_calls"):
            if k in o:
                argv += ["--" + k, str(o[k])]
        if o.get("cmdline"):
            argv += ["--cmdline", str(tmp_path / "CMD")]
        if o.get("ref_fn"):
            argv += ["--ref_fn", str(tmp_path / "ref.fa")]
        postprocess_vcf_main(argv)
        assert out.read_text() == case["out"], o


def test_bench_rejects_rank_count_mismatch():
    """--gpus must agree with the number of ranks the launcher started (the round-1 bench silently ignored --gpus)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_platform_table_matches_reference_names():
    """Every platform name of the reference resolves to the reference's min_bq and family (shared/param.py:34,
    run_clairs_to:590-595, 910-918, 1089-1096); unknown names are refused, never mapped to ont."""
    import pytest
    from clairs_to_amd.platforms import resolve_platform
    assert resolve_platform("hifi_revio") == ("hifi_revio", "hifi", 0)
    assert resolve_platform("ont_r10_dorado_hac_4khz") == ("ont_r10_dorado_hac_4khz", "ont", 15)
    assert resolve_platform("ont_r10_guppy_5khz")[2] == 15 and resolve_platform("ont_r10_guppy_hac_5khz")[2] == 15
    assert resolve_platform("r1041_e82_400bps_sup_v420") == ("ont_r10_dorado_sup_5khz", "ont", 20)
    assert resolve_platform("ilmn_ssrs") == ("ilmn_ssrs", "ilmn", 0)
    for name in ("ont", "ilmn", "hifi", "hifi_ss", "ont_r10_dorado_sup_5khz_ssrs"):
        resolve_platform(name)
    with pytest.raises(SystemExit):
        resolve_platform("pacbio")
    with pytest.raises(ValueError):
        resolve_platform("", exit_on_unknown=False)


def test_c_bed_reader_equals_the_row_loop(tmp_path):
    """cto_bed_centres (one C call per chunk file) against read_candidates, the row-by-row restatement of
    create_tensor_pileup_calling.py:347-370: windows clipped at the contig start, other contigs, duplicates, a type column"""
    from clairs_to_amd.create_tensor_pileup_calling import read_candidates, read_candidate_positions
    rng = np.random.default_rng(4)
    rows = []
    for x in sorted(set(rng.integers(1, 5000, size=700).tolist()) | {1, 2, 17, 18, 19}):
        rows.append("chr2\t%d\t%d" % (max(x - 17, 1) if x % 3 else x - 17, x + 17))
    rows += ["chr20\t100\t134", "chr2\t40\t74", "chr2\t40\t74\tsnv", "", "chr2\t7"]
    bed = tmp_path / "c.bed"
    bed.write_text("\n".join(rows) + "\n")
    centres, s, e = read_candidates(str(bed), "chr2")
    pos, s2, e2 = read_candidate_positions(str(bed), "chr2")
    assert pos.dtype == np.int32 and pos.tolist() == sorted(centres) and (s, e) == (s2, e2)
    pos, s2, e2 = read_candidate_positions(str(bed), "chrZ")
    assert len(pos) == 0 and e2 == 0
    bad = tmp_path / "bad.bed"
    bad.write_text("chr2\tx\t5\n")
    from clairs_to_amd._lib import CtoError
    with pytest.raises(CtoError):
        read_candidate_positions(str(bad), "chr2")


def test_short_read_platforms_warn_about_the_unpinned_bam_reader(capsys):
    """--bam_reader native | gpu on an Illumina platform: a loud warning (the mate-overlap rule of csrc/bam.cpp is not pinned against
    samtools), once per process; long-read platforms and the samtools producer stay quiet."""
    from clairs_to_amd import platforms
    platforms._warned.clear()
    platforms.warn_unpinned_bam_reader("ont_r10_dorado_sup_5khz", "native")
    platforms.warn_unpinned_bam_reader("hifi_revio", "gpu")
    platforms.warn_unpinned_bam_reader("ilmn", "samtools")
    assert capsys.readouterr().err == ""
    platforms.warn_unpinned_bam_reader("ilmn_ss", "native")
    platforms.warn_unpinned_bam_reader("ilmn", "gpu")
    err = capsys.readouterr().err
    assert err.count("[WARNING]") == 1 and "mate-overlap" in err and "NOT pinned" in err


def test_split_operands_is_an_opt_in_of_the_drivers_and_of_a_module():
    """The experimental split-operand kernels are never the default: the drivers take --split_operands f16|bf16 (nothing else), a
    module carries the choice as an attribute, and changing it changes the key under which its C-ABI handle is cached (a handle
    is rebuilt, not reused, when the arithmetic changes)."""
    from argparse import ArgumentParser
    from clairs_to_amd import nn_shims
    from clairs_to_amd.pileup_call import add_common_arguments
    p = ArgumentParser()
    add_common_arguments(p)
    base = ["--tumor_bam_fn", "t.bam", "--ref_fn", "r.fa", "--ctg_name", "c", "--chkpnt_fn_acgt", "a", "--chkpnt_fn_nacgt", "n",
            "--likelihood_matrix_data", "l", "--platform", "ont_r10_dorado_sup_5khz"]
    known = {a.dest for a in p._actions}
    argv = [x for i, x in enumerate(base) if (x.startswith("--") and x[2:] in known) or (i and base[i - 1].startswith("--") and base[i - 1][2:] in known and not x.startswith("--"))]
    assert p.parse_args(argv).split_operands is None
    assert p.parse_args(argv + ["--split_operands", "f16"]).split_operands == "f16"
    with pytest.raises(SystemExit):
        p.parse_args(argv + ["--split_operands", "fp8"])
    m = nn_shims.BiGRU_NACGT()
    assert m.split_operands is None
    v0 = m._weights_version()
    m.split_operands = "f16"
    assert m._weights_version() != v0 and m._weights_version()[0] == "f16"
