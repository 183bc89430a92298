"""`python -m clairs_to_amd <submodule> ...` - same dispatch style as the reference's clairs_to.py:84-107 for the
hot-path sub-modules this package replaces; each takes the argv run_clairs_to builds for its namesake (tests/test_cli_argv.py)."""
import importlib
import sys

SUBMODULES = ("extract_candidates_calling", "concat_files", "create_tensor_pileup_calling", "predict", "call_variants", "pileup_call", "call_chunks",
              "sort_vcf", "postprocess_vcf", "haplotype_filtering", "realign_reads", "realign_variants")


def dispatch(name, argv):
    """Run sub-module `name` on `argv` (a list, without the program and sub-module names) in this process."""
    if name in ("sort_vcf", "postprocess_vcf"):          # the host-side tail lives in one module
        mod = importlib.import_module("clairs_to_amd.postprocess_vcf")
        return getattr(mod, name + "_main")(argv)
    return importlib.import_module("clairs_to_amd." + name).main(argv)


def main():
    if len(sys.argv) < 2 or sys.argv[1] not in SUBMODULES:
        sys.exit("usage: python -m clairs_to_amd {%s} [options]" % "|".join(SUBMODULES))
    name = sys.argv.pop(1)
    dispatch(name, sys.argv[1:])


if __name__ == "__main__":
    main()
