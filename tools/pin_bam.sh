#!/usr/bin/env bash
# One command, exit code = verdict (0 equal / 1 different / 2 no samtools here): see tools/pin_bam.py.
#   SAMTOOLS=/path/to/samtools bash tools/pin_bam.sh
cd "$(dirname "$0")/.."
[ -f clairs_to_amd/libclairsto_amd.so ] || python -c "import __graft_entry__ as g; g.build()" || exit 2
exec python tools/pin_bam.py --samtools "${SAMTOOLS:-samtools}" "$@"
