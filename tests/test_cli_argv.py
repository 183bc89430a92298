"""The command-line seam takes the reference's argv verbatim.

`tests/golden/argv.json.gz` (made by gen_cli.py from the reference in the build container) holds (a) the option table of every reference
sub-module this package mirrors and (b) the argv lists `run_clairs_to --dry_run` builds for ont / ilmn / hifi runs (SNV only, SNV + indel,
--print_ref_calls, --bed_fn, --call_indels_only_in_these_regions, --genotyping_mode_vcf_fn, --hybrid_mode_vcf_fn, --debug ...).  Every mirror's
parser must accept every one of its reference's options with the reference's arity, and every captured argv as it stands (GNU parallel's
replacement strings filled in the way parallel fills them).  No GPU needed: parsers only."""
import importlib

import pytest

from conftest import load_json_gz

PARSERS = {
    "extract_candidates_calling": ("clairs_to_amd.extract_candidates_calling", "build_parser"),
    "concat_files": ("clairs_to_amd.concat_files", "build_parser"),
    "create_tensor_pileup_calling": ("clairs_to_amd.create_tensor_pileup_calling", "build_parser"),
    "predict": ("clairs_to_amd.predict", "build_parser"),
    "call_variants": ("clairs_to_amd.call_variants", "build_parser"),
    "sort_vcf": ("clairs_to_amd.postprocess_vcf", "build_sort_vcf_parser"),
    "postprocess_vcf": ("clairs_to_amd.postprocess_vcf", "build_postprocess_vcf_parser"),
    "haplotype_filtering": ("clairs_to_amd.haplotype_filtering", "build_parser"),
    "realign_reads": ("clairs_to_amd.realign_reads", "build_parser"),
    "realign_variants": ("clairs_to_amd.realign_variants", "build_parser"),
}
# sub-modules of run_clairs_to's command list that SURVEY.md section 8 leaves outside the hot path (no mirror exists; the reference's stay in place)
OUT_OF_SCOPE = {"nonsomatic_tagging", "select_hetero_snp_for_phasing", "postfilter_variants", "add_back_missing_variants_in_genotyping",
                "cna_germline_tagging"}
SAMPLE = {"int": "7", "float": "0.25", "str": "x", "str2bool": "True", "str_none": "None", None: "x"}
# what the mirrors additionally insist on (their inputs), so that a one-option command line parses at all
REQUIRED = {
    "extract_candidates_calling": ["--candidates_folder", "c", "--ref_fn", "r.fa", "--ctg_name", "chr20"],
    "concat_files": ["--input_dir", "d", "--input_prefix", "p"],
    "create_tensor_pileup_calling": ["--ref_fn", "r.fa", "--ctg_name", "chr20", "--candidates_bed_regions", "b"],
    "predict": ["--tensor_fn_acgt", "a", "--tensor_fn_nacgt", "n", "--chkpnt_fn_acgt", "a.pkl", "--chkpnt_fn_nacgt", "n.pkl"],
    "call_variants": ["--call_fn", "o.vcf", "--predict_fn", "p", "--likelihood_matrix_data", "l.txt"],
    "sort_vcf": ["--output_fn", "o.vcf", "--input_dir", "d", "--contigs_fn", "C"],
    "postprocess_vcf": ["--output_fn", "o.vcf", "--pileup_vcf_fn", "i.vcf"],
}


def parser_of(sub):
    mod, fn = PARSERS[sub]
    return getattr(importlib.import_module(mod), fn)()


def parse(sub, argv):
    try:
        return parser_of(sub).parse_args(argv)
    except SystemExit as e:                            # argparse reports through sys.exit(2)
        pytest.fail("%s rejects %r (exit %s)" % (sub, argv, e.code))


@pytest.fixture(scope="module")
def golden():
    return load_json_gz("argv.json.gz")


@pytest.mark.parametrize("sub", sorted(PARSERS))
def test_every_reference_option_is_accepted(golden, sub):
    table = golden["parsers"][sub]
    assert len(table) >= 6
    base = REQUIRED.get(sub, [])
    for opt in table:
        for name in opt["options"]:
            if name in base:
                continue
            if opt["action"] == "_StoreTrueAction":
                argv = [name]
            elif opt["nargs"] == "?":
                parse(sub, base + [name])                          # the bare form too
                argv = [name, SAMPLE[opt["type"]]]
            else:
                argv = [name, SAMPLE[opt["type"]]]
            parse(sub, base + argv)
    # and all of them on one command line
    argv = list(base)
    for opt in table:
        name = opt["options"][-1]
        if name in argv:
            continue
        argv += [name] if opt["action"] == "_StoreTrueAction" else [name, SAMPLE[opt["type"]]]
    parse(sub, argv)


def fill(argv, source):
    """GNU parallel's replacement strings for one representative input row"""
    fields = ["chr20", "2", "3"] if source == "CHUNK_LIST" else (["chr20"] if source == "CONTIGS" else ["/w/tmp/candidates/chr20.1_0_1_snv"])
    out = []
    for t in argv:
        for n, v in enumerate(fields, 1):
            t = t.replace("{%d}" % n, v)
        base = fields[0].rsplit("/", 1)[-1]
        t = t.replace("{1/.}", base.rsplit(".", 1)[0] if "." in base else base).replace("{1/}", base)
        out.append(t.replace("@W@", "/w").replace("@T@", "/t").replace("@REF@", "/ref"))
    return out


def test_every_captured_argv_parses(golden):
    seen, n = set(), 0
    for run in golden["runs"]:
        for inv in run["invocations"]:
            sub = inv["submodule"]
            seen.add(sub)
            if sub in OUT_OF_SCOPE:
                continue
            assert sub in PARSERS, "run_clairs_to runs %s: neither mirrored nor listed as out of scope" % sub
            argv = fill(inv["argv"], inv["source"])
            assert not any("{" in t and "}" in t for t in argv), argv
            a = parse(sub, argv)
            n += 1
            # spot checks that the values land where the mirrors read them
            if sub == "call_variants":
                assert a.call_fn.endswith(".vcf") and a.disable_indel_calling in (True, False)
                assert (a.ref_fn is not None) == a.disable_indel_calling          # the SNV command carries --ref_fn, the indel one does not
            if sub == "extract_candidates_calling":
                assert a.chunk_id == 2 and a.chunk_num == 3 and a.bed_fn.endswith("split_beds/chr20")
                assert a.genotyping_mode_vcf_fn is None or a.genotyping_mode_vcf_fn.endswith(".vcf")
            if sub == "postprocess_vcf":
                assert a.cmdline.endswith("/CMD")
            if sub == "predict":
                assert a.pileup is True
    assert n > 150
    assert set(PARSERS) - {"realign_reads"} <= seen                # realign_reads is run by realign_variants, not by run_clairs_to
    assert seen - set(PARSERS) <= OUT_OF_SCOPE          # (cna_germline_tagging needs Verdict's resource files even for a dry run: not in the fixture)


def test_unsupported_modes_end_the_run_loudly():
    from clairs_to_amd import _cli
    p = parser_of("extract_candidates_calling")
    a = p.parse_args(REQUIRED["extract_candidates_calling"] + ["--truth_vcf_fn", "t.vcf"])
    with pytest.raises(SystemExit) as e:
        _cli.check_unsupported(p, a)
    assert "--truth_vcf_fn" in str(e.value)
    a = p.parse_args(REQUIRED["extract_candidates_calling"] + ["--store_tumor_infos", "False"])
    _cli.check_unsupported(p, a)                                    # the default value is no request
    with pytest.raises(SystemExit):
        p.parse_args(REQUIRED["extract_candidates_calling"] + ["--select_indel_candidates", "maybe"])      # str2bool of the reference


def test_vcf_header_is_the_references(tmp_path):
    """The header of a chunk VCF, byte for byte: cli_run.json.gz holds whole p_<chunk>.vcf files the reference's call_variants wrote - the SNV
    ones with the ##contig lines of --ref_fn's .fai, the indel ones without (run_clairs_to:1300 vs :1631-1645)."""
    from clairs_to_amd.call_variants import chunk_vcf_header, vcf_header
    g = load_json_gz("cli_run.json.gz")
    run = g["executed"]["ont"]
    fa = tmp_path / "ref.fa"
    (tmp_path / "ref.fa.fai").write_text(open_fai(g))
    for name, text in run["vcf_output"].items():
        head = "".join(r + "\n" for r in text.split("\n") if r.startswith("#"))
        K = 4 if name.endswith("_snv.vcf") else 6
        assert chunk_vcf_header(str(fa), K) == head, name
    assert "##contig=<ID=chr20," in run["vcf_output"]["p_chr20.0_0_1_snv.vcf"] and "##contig" not in run["vcf_output"]["p_chr20.0_0_1_indel.vcf"]
    h = vcf_header(str(fa), "chr21", "S1", cmdline="run_clairs_to -T t.bam")
    rows = h.split("\n")
    assert rows[3] == "##cmdline=run_clairs_to -T t.bam" and rows[-3].startswith("##contig=<ID=chr21,") and rows[-2].endswith("\tS1")
    assert sum(r.startswith("##contig") for r in rows) == 1


def open_fai(g):
    """the .fai clisim.write_inputs writes for the fixture's reference (not stored: recomputed from the FASTA text)"""
    out, name, n, off, pos = [], None, 0, 0, 0
    for line in g["inputs"]["ref"].split("\n"):
        if line.startswith(">"):
            if name:
                out.append((name, n, off))
            name, n = line[1:], 0
            off = pos + len(line) + 1
        else:
            n += len(line)
        pos += len(line) + 1
    out.append((name, n, off))
    return "".join("%s\t%d\t%d\t60\t61\n" % r for r in out)
