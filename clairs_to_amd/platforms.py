"""Platform names of the reference and what the hot path derives from them.

The reference resolves `--min_bq` from the FULL platform name (shared/param.py:34 min_bq_dict, run_clairs_to:910-911),
translates ONT basecaller model names first (shared/param.py:9-15, run_clairs_to:590-595), rejects names it does not know
(run_clairs_to:913-918) and only then collapses the name to its family ('ont' / 'hifi' / 'ilmn', run_clairs_to:1089-1096),
which is what the sub-commands and the QUAL threshold tables (shared/param.py:35-40) see.  For the 'ilmn' family the NEG
tensors are a symlink to the AFF tensors (run_clairs_to:1248-1252, 1587-1592): the NEG network reads the AFF pass.
"""
import sys

MODEL_NAME_TO_PLATFORM = {
    "r1041_e82_400bps_sup_v420": "ont_r10_dorado_sup_5khz",
    "r1041_e82_400bps_sup_v410": "ont_r10_dorado_sup_4khz",
    "r1041_e82_400bps_hac_v410": "ont_r10_dorado_hac_4khz",
    "r1041_e82_400bps_sup_g615": "ont_r10_guppy_sup_4khz",
    "r1041_e82_400bps_hac_g657": "ont_r10_guppy_hac_5khz",
}

_Q20 = ("ont", "ont_r10_dorado_sup_4khz", "ont_r10_dorado_sup_5khz", "ont_r10_dorado_sup_5khz_ss", "ont_r10_dorado_sup_5khz_ssrs",
        "ont_r10_guppy_sup_4khz", "ont_r10_dorado_4khz", "ont_r10_dorado_5khz", "ont_r10_guppy", "ont_r10_guppy_4khz")
_Q15 = ("ont_r10_dorado_hac_4khz", "ont_r10_guppy_hac_5khz", "ont_r10_guppy_5khz")
_Q0 = ("ilmn", "ilmn_ss", "ilmn_ssrs", "hifi", "hifi_ss", "hifi_ssrs", "hifi_revio", "hifi_revio_ss", "hifi_revio_ssrs")
MIN_BQ = {**{p: 20 for p in _Q20}, **{p: 15 for p in _Q15}, **{p: 0 for p in _Q0}}


def family_of(platform):
    """'ont' / 'hifi' / 'ilmn' (run_clairs_to:1089-1096)."""
    for fam in ("ont", "hifi", "ilmn"):
        if platform.startswith(fam):
            return fam
    return None


def resolve_platform(platform, exit_on_unknown=True):
    """name (platform or ONT model name) -> (canonical name, family, default AFF-pass min_bq).  Unknown names end the run, as in
    the reference; they never fall back to another platform's gates."""
    name = MODEL_NAME_TO_PLATFORM.get(platform, platform)
    if name not in MIN_BQ:
        msg = "[ERROR] Invalid platform input '%s', optional: {%s}" % (platform, ", ".join(sorted(MIN_BQ)))
        if exit_on_unknown:
            sys.exit(msg)
        raise ValueError(msg)
    return name, family_of(name), MIN_BQ[name]


_warned = set()


def warn_unpinned_bam_reader(platform, reader):
    """The built-in BAM readers (`--bam_reader native | gpu`) are parity-UNPINNED against samtools (absent from the build's boxes).
    For unpaired long reads the rules that matter are the SAM specification's; for PAIRED short reads the pileup also depends on
    htslib's mate-overlap handling (which mate keeps its base quality where the two overlap - recent htslib versions differ here),
    which csrc/bam.cpp restates from reading, not from a run.  Short-read platforms therefore get a loud warning, once per process;
    `samtools` stays the default producer."""
    if reader in ("native", "gpu") and family_of(str(platform)) == "ilmn" and "ilmn" not in _warned:
        _warned.add("ilmn")
        sys.stderr.write("[WARNING] --bam_reader %s on a short-read platform (%s): the built-in BAM reader's mate-overlap rule is NOT pinned "
                         "against samtools/htslib (csrc/bam.cpp); overlapping read pairs may pile up differently than `samtools mpileup` "
                         "would print them.  Use --bam_reader samtools for paired-end data unless you have checked this reader against "
                         "your samtools version.\n" % (reader, platform))
        sys.stderr.flush()
