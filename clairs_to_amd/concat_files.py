"""`clairs_to.py concat_files` counterpart (reference: src/concat_files.py:37-66): the second half of run_clairs_to's STEP 1 command (:1221-1225) and
the first of STEP 6 (:1564-1567) - the per-chunk `SNV_CANDIDATES_FILE_<ctg>_<chunk>` / `INDEL_CANDIDATES_FILE_...` lists that
`extract_candidates_calling` leaves in the candidates folder become the one list GNU parallel feeds STEP 2 / STEP 6 from."""
import os
import sys
from argparse import ArgumentParser


def concat_files(input_dir, input_prefix, output_fn=None, output_dir=None, is_snv=False, is_indel=False):
    output_dir = output_dir if output_dir is not None else input_dir
    if not os.path.exists(input_dir):
        sys.exit("[ERROR] The input prefix is not found: {}".format(input_prefix))
    if output_fn is not None and "/" not in output_fn:
        output_fn = os.path.join(output_dir, output_fn)
    if output_fn is None and is_snv:
        output_fn = os.path.join(output_dir, "SNV_CANDIDATES_FILES")
    elif output_fn is None and is_indel:
        output_fn = os.path.join(output_dir, "INDEL_CANDIDATES_FILES")
    if output_fn is None:
        sys.exit("[ERROR] concat_files needs --output_fn, --is_snv or --is_indel")
    # the output shares the prefix of its inputs when it is re-made in place: list first, then write
    names = [f for f in os.listdir(input_dir) if f.startswith(input_prefix) and os.path.join(input_dir, f) != output_fn]
    rows = []
    for name in names:
        with open(os.path.join(input_dir, name)) as f:
            rows += [row for row in f if row.rstrip() != ""]
    with open(output_fn, "w") as out:
        out.writelines(rows)
    return len(rows)


def build_parser():
    p = ArgumentParser(description="Concat file with the same input prefix")
    p.add_argument("--input_dir", type=str, default=None, required=True)
    p.add_argument("--output_dir", type=str, default=None)
    p.add_argument("--input_prefix", type=str, default=None, required=True)
    p.add_argument("--output_fn", type=str, default=None)
    p.add_argument("--is_snv", action="store_true")
    p.add_argument("--is_indel", action="store_true")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    concat_files(a.input_dir, a.input_prefix, a.output_fn, a.output_dir, a.is_snv, a.is_indel)


if __name__ == "__main__":
    main()
