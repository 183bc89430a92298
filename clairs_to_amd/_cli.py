"""What the command-line mirrors share: the reference's argparse value types (shared/utils.py str2bool / str_none), and the rule for the
options of a reference sub-module that this package does not act on.

`run_clairs_to` builds a FIXED argv per sub-module (run_clairs_to:1194-1226 STEP 1, :1273-1308 STEP 2, :1487-1529 STEP 4/5, :1562-1647 STEP 6),
so a mirror must take every option the reference's parser takes - `tests/golden/argv.json.gz` holds those parsers' option tables and the
argv lists of `run_clairs_to --dry_run`, and `tests/test_cli_argv.py` feeds them to the parsers built here.  An option falls in one of three classes:
  * acted on - declared by the mirror itself;
  * `ignored`: options the reference's own sub-module never reads on the calling path (`--ref_fn` / `--samtools` of call_variants, `--pypy3`,
    `--parallel`, `--debug` ...) or that only choose how it spends host threads: parsed with the reference's arity, then dropped;
  * `unsupported`: options that switch the reference's sub-module into a mode outside the calling path (training-set helpers: `--truth_vcf_fn`,
    `--alt_fn`, `--store_tumor_infos` ...): parsed, and the run ends with a message naming the option when it carries a value other than its
    default - never silently ignored."""
import argparse
import sys


def str2bool(v):
    """shared/utils.py str2bool: the spellings the reference accepts, an error for anything else"""
    if isinstance(v, bool):
        return v
    s = str(v).lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def str_none(v):
    """shared/utils.py str_none: the string 'None' (what `str(None)` puts on run_clairs_to's command lines) is no value"""
    if v is None:
        return None
    return None if str(v).upper() == "NONE" else v


KINDS = {"str": str, "int": int, "float": float, "bool": str2bool, "str_none": str_none}


def add_ignored(parser, **options):
    """options: name -> 'str' | 'int' | 'float' | 'bool' | 'str_none' (one value, the reference's type) or 'flag' (store_true)"""
    g = parser.add_argument_group("accepted as the reference accepts them; not used here")
    for name, kind in options.items():
        if kind == "flag":
            g.add_argument("--" + name, action="store_true", help=argparse.SUPPRESS)
        else:
            g.add_argument("--" + name, type=KINDS[kind], default=None, help=argparse.SUPPRESS)


def add_unsupported(parser, **options):
    """options: name -> (kind, default).  check_unsupported() ends the run when one of them was given another value."""
    g = parser.add_argument_group("modes of the reference's sub-module outside the calling path: rejected when set")
    table = getattr(parser, "_cto_unsupported", {})
    for name, (kind, default) in options.items():
        if kind == "flag":
            g.add_argument("--" + name, action="store_true", help=argparse.SUPPRESS)
            default = False
        elif kind.endswith("?"):                  # the reference declares a few options with nargs='?'
            g.add_argument("--" + name, nargs="?", type=KINDS[kind[:-1]], default=default, help=argparse.SUPPRESS)
        else:
            g.add_argument("--" + name, type=KINDS[kind], default=default, help=argparse.SUPPRESS)
        table[name] = default
    parser._cto_unsupported = table


def check_unsupported(parser, args):
    for name, default in getattr(parser, "_cto_unsupported", {}).items():
        v = getattr(args, name)
        if v != default and v is not None:
            sys.exit("[ERROR] --{} {}: this mode of the reference's sub-module is not part of the calling path and is not implemented by "
                     "clairs_to_amd (it would be ignored silently otherwise)".format(name, v))
