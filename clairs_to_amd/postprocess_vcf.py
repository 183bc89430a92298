"""Host-side tail of the pileup calling path: merge the per-chunk VCFs, then apply the QUAL / AF gates.

Mirrors, for the records the hot path produces, what the reference does after `call_variants`:
  * `sort_vcf`        - src/sort_vcf.py:129-226 (`sort_vcf_from`): concatenate `<prefix>*<suffix>` chunk files of every contig
                        in contig order (chr1..22,X,Y / 1..22,X,Y first, then the order of the contigs file), records
                        sorted by position, later files overriding earlier ones at the same position, first-seen header lines;
  * `postprocess_vcf` - src/postprocess_vcf.py:61-195 (`mark_low_qual`, `update_GQ`, `merge_vcf`): drop PASS records under the
                        AF cut-off (and, optionally, under `max_qual_filter_pileup_calls`), re-derive GQ from QUAL, zero the
                        QUAL of non-PASS records other than RefCall / NonSomatic, mark LowQual by the platform's thresholds
                        (phaseable `H` records have their own), write the header up to the last FORMAT line + contigs.
Pure text processing: no device work, no oracle; parity is pinned by tests/golden/post.json.gz (reference outputs).
"""
import os
from argparse import ArgumentParser

MAJOR_CONTIGS = (["chr" + str(a) for a in list(range(1, 23)) + ["X", "Y"]] +
                 [str(a) for a in list(range(1, 23)) + ["X", "Y"]])

# shared/param.py:35-37, 48
MIN_QUAL = {"ont": 8, "ilmn": 4, "hifi": 8, "hifi_revio": 8}
MIN_QUAL_PHASEABLE = {"ont": 8, "ilmn": 4, "hifi": 8, "hifi_revio": 8}
MIN_QUAL_UNPHASEABLE = {"ont": 12, "ilmn": 6, "hifi": 12, "hifi_revio": 12}
MIN_AF = {"ont": 0.05, "ilmn": 0.05, "hifi": 0.05}

LAST_FORMAT_LINE = '##FORMAT=<ID=TU,Number=1,Type=Integer,Description="Count of T in the tumor BAM">'


def contig_order(names):
    """Stable sort of `names` by the reference's rule: position of the first match in MAJOR_CONTIGS + names."""
    rank = {}
    for i, n in enumerate(list(MAJOR_CONTIGS) + list(names)):
        rank.setdefault(n, i)
    return sorted(names, key=lambda n: rank[n])


def _header_from_fai(ref_fn, sample_name, only=None):
    from .call_variants import VCF_HEADER
    out = VCF_HEADER
    if ref_fn is not None:
        for row in open(ref_fn + ".fai"):
            c = row.strip().split("\t")
            if only is None or c[0] in only:
                out += "##contig=<ID=%s,length=%s>\n" % (c[0], c[1])
    return out + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s" % sample_name


def sort_vcf(input_dir, output_fn, contigs, vcf_fn_prefix=None, vcf_fn_suffix=".vcf", ref_fn=None, sample_name="SAMPLE",
             only_files=None):
    """Merge chunk VCFs; returns the number of records written.  only_files (not in the reference, which merges whatever the
    directory holds): restrict the merge to these file names - call_chunks passes the p_<chunk>.vcf names of its own chunk
    list so that leftovers of an earlier run in the same directory stay out."""
    files = os.listdir(input_dir)
    if only_files is not None:
        keep = set(only_files)
        files = [f for f in files if f in keep]
    if vcf_fn_prefix is not None:
        files = [f for f in files if f.startswith(vcf_fn_prefix)]
    if vcf_fn_suffix is not None:
        files = [f for f in files if f.endswith(vcf_fn_suffix)]
    header, n_rows, n_records = [], 0, 0
    seen = set()
    body = []
    written_header = None        # the reference emits the header lines collected up to the first contig that has any
    for ctg in contig_order(list(contigs)):
        at = {}
        for fn in (f for f in files if ctg in f):          # file names carry the contig (p_<ctg>.<chunk>...)
            with open(os.path.join(input_dir, fn)) as fh:
                for row in fh:
                    n_rows += 1
                    if row.startswith("#"):
                        if row not in seen:
                            seen.add(row)
                            header.append(row)
                        continue
                    name, pos = row.split(maxsplit=2)[:2]
                    if name != ctg:                         # chr1 also matches chr11's files: leave them to their contig
                        break
                    at[int(pos)] = row
        if written_header is None and header:
            written_header = list(header)
        body.extend(at[p] for p in sorted(at))
        n_records += len(at)
    with open(output_fn, "w") as out:
        if n_records == 0:                                  # the reference rewrites the file as a bare header (no newline)
            out.write(_header_from_fai(ref_fn, sample_name))
        else:
            out.write("".join(written_header or []))
            out.write("".join(body))
    return n_records


def _with_gq(cols):
    """GQ := int(QUAL) when QUAL > 0, else int(float(old GQ)) (postprocess_vcf.py:85-91)."""
    keys = cols[8].split(":")
    vals = cols[9].split(":")
    i = keys.index("GQ")
    q = float(cols[5])
    vals[i] = str(int(q)) if q > 0.0 else str(int(float(vals[i])))
    cols[9] = ":".join(vals)
    return cols


def _mark_low_qual(row, platform, q_all, q_phaseable, q_unphaseable):
    if row == "" or "RefCall" in row or "LowQual" in row:
        return row
    cols = row.split("\t")
    qual = float(cols[5])
    if q_all and qual < float(q_all):
        if "NonSomatic" in row:
            cols[6], cols[5] = "LowQual;NonSomatic", "0.0000"
        else:
            cols[6] = "LowQual"
    if platform != "ilmn" and "PASS" in row:
        cut = q_phaseable if "H" in cols[7].split(";") else q_unphaseable
        if cut and qual < float(cut):
            cols[6] = "LowQual"
    return "\t".join(cols)


def postprocess_vcf(pileup_vcf_fn, output_fn, platform="ont", qual=None, qual_cutoff_phaseable_region=None,
                    qual_cutoff_unphaseable_region=None, af=None, max_qual_filter_pileup_calls=None, ref_fn=None,
                    sample_name="SAMPLE", cmdline=None):
    """Returns the number of records written."""
    q_all = qual if qual is not None else MIN_QUAL[platform]
    q_ph = qual_cutoff_phaseable_region if qual_cutoff_phaseable_region is not None else MIN_QUAL_PHASEABLE[platform]
    q_un = qual_cutoff_unphaseable_region if qual_cutoff_unphaseable_region is not None else MIN_QUAL_UNPHASEABLE[platform]
    af_cut = af if af is not None else MIN_AF[platform]
    header_lines = []
    kept = {}                     # contig -> {pos: row}; insertion order of contigs = first PASS record kept
    parked = {}                   # (contig, pos) -> original row of non-PASS records, re-attached afterwards
    for row in open(pileup_vcf_fn):
        if row.startswith("#"):
            header_lines.append(row)
            continue
        cols = row.strip().split()
        ctg, pos, q = cols[0], int(cols[1]), float(cols[5])
        if cols[6] != "PASS":
            parked[(ctg, pos)] = row
            continue
        if max_qual_filter_pileup_calls is not None:
            if q < float(max_qual_filter_pileup_calls):
                continue
            if platform == "ilmn":                            # kept even when the AF gate below drops it (reference behaviour)
                kept.setdefault(ctg, {})[pos] = "\t".join(_with_gq(list(cols))) + "\n"
                cols = row.strip().split()
        if af_cut is not None:
            keys = cols[8].split(":")
            i = keys.index("AF") if "AF" in keys else keys.index("VAF")
            if float(cols[9].split(":")[i]) < af_cut:
                continue
        kept.setdefault(ctg, {})[pos] = "\t".join(_with_gq(cols)) + "\n"
    for (ctg, pos), row in parked.items():
        if ctg in kept and pos in kept[ctg]:
            continue
        cols = row.strip().split()
        if cols[6] not in ("NonSomatic", "RefCall"):
            cols[5] = "0.0000"
        kept.setdefault(ctg, {})[pos] = "\t".join(_with_gq(cols)) + "\n"

    # header: the input's, cut after the last FORMAT line; optional ##cmdline; contigs of the output; column line
    head = []
    for line in "".join(header_lines).split("\n"):
        head.append(line)
        if LAST_FORMAT_LINE in line:
            break
    else:
        head = head[:1]           # the reference keeps only the first line when the marker is absent
    if cmdline:
        head.insert(3 if len(head) >= 3 else len(head) - 1, "##cmdline=%s" % cmdline)
    text = "\n".join(head) + "\n"
    if ref_fn is not None:
        names = set(kept)
        for row in open(ref_fn + ".fai"):
            c = row.strip().split("\t")
            if c[0] in names:
                text += "##contig=<ID=%s,length=%s>\n" % (c[0], c[1])
    text += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n" % sample_name
    n = 0
    with open(output_fn, "w") as out:
        out.write(text)
        for ctg in contig_order(list(kept)):
            for pos in sorted(kept[ctg]):
                out.write(_mark_low_qual(kept[ctg][pos], platform, q_all, q_ph, q_un))
                n += 1
    return n


def _none_or_float(v):
    return None if v is None or str(v) == "None" else float(v)


def compress_index_vcf(vcf_fn):
    """src/postprocess_vcf.py:54-59: `bgzip -f` then `tabix -f -p vcf`, run the same way - through the shell, outcome not checked, so
    a host without the two tools keeps the plain .vcf exactly as the reference would leave it"""
    import subprocess
    subprocess.run("bgzip -f {}".format(vcf_fn), shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    subprocess.run("tabix -f -p vcf {}.gz".format(vcf_fn), shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def build_sort_vcf_parser():
    from ._cli import str2bool
    ap = ArgumentParser(description="merge and sort chunk VCFs (mirror of src/sort_vcf.py)")
    ap.add_argument("--output_fn", required=True)
    ap.add_argument("--input_dir", required=True)
    ap.add_argument("--vcf_fn_prefix", default=None)
    ap.add_argument("--vcf_fn_suffix", default=".vcf")
    ap.add_argument("--ref_fn", default=None)
    ap.add_argument("--sample_name", default="SAMPLE")
    ap.add_argument("--contigs_fn", required=True)
    ap.add_argument("--compress_vcf", type=str2bool, default=False)          # src/sort_vcf.py:248
    return ap


def sort_vcf_main(argv=None):
    a = build_sort_vcf_parser().parse_args(argv)
    contigs = [r.rstrip() for r in open(a.contigs_fn)]
    n = sort_vcf(a.input_dir, a.output_fn, contigs, a.vcf_fn_prefix, a.vcf_fn_suffix, a.ref_fn, a.sample_name)
    if a.compress_vcf:
        compress_index_vcf(a.output_fn)
    return n


def build_postprocess_vcf_parser():
    from ._cli import add_ignored, add_unsupported, str2bool, str_none
    ap = ArgumentParser(description="QUAL / AF gates on the merged pileup VCF (mirror of src/postprocess_vcf.py)")
    ap.add_argument("--platform", default="ont")
    ap.add_argument("--output_fn", required=True)
    ap.add_argument("--pileup_vcf_fn", required=True)
    ap.add_argument("--ref_fn", default=None)
    ap.add_argument("--sample_name", default="SAMPLE")
    ap.add_argument("--cmdline", type=str_none, default=None)
    ap.add_argument("--qual", default=None)
    ap.add_argument("--qual_cutoff_phaseable_region", default=None)
    ap.add_argument("--qual_cutoff_unphaseable_region", default=None)
    ap.add_argument("--af", default=None)
    ap.add_argument("--max_qual_filter_pileup_calls", default=None)
    ap.add_argument("--compress_vcf", type=str2bool, default=True)           # src/postprocess_vcf.py:223: bgzip + tabix when they exist
    # src/postprocess_vcf.py:239-256: declared there, read nowhere (run_clairs_to:1527, 1783-1784 passes the first two on every run)
    add_ignored(ap, disable_indel_calling="bool", indel_calling="flag", bed_format="flag", prefer_recall="bool")
    return ap


def postprocess_vcf_main(argv=None):
    a = build_postprocess_vcf_parser().parse_args(argv)
    cmd = None
    if a.cmdline is not None and os.path.exists(a.cmdline):
        cmd = open(a.cmdline).read().rstrip()
    n = postprocess_vcf(a.pileup_vcf_fn, a.output_fn, a.platform, _none_or_float(a.qual),
                        _none_or_float(a.qual_cutoff_phaseable_region), _none_or_float(a.qual_cutoff_unphaseable_region),
                        _none_or_float(a.af), _none_or_float(a.max_qual_filter_pileup_calls), a.ref_fn, a.sample_name, cmd)
    if a.compress_vcf:
        compress_index_vcf(a.output_fn)
    return n
