#!/usr/bin/env bash
# Pin-on-arrival for the de Bruijn consensus (SURVEY.md 8f #4b; DESIGN.md section 0: "parity unpinned").
#
# Needs: Boost headers (boost/graph/adjacency_list.hpp - header-only parts suffice) and the reference checkout.
#   REFERENCE=/root/reference BOOST_INC=/usr/include bash tools/pin_dbg.sh
# Builds the reference's own src/realign/debruijn_graph.cpp, from where it lies, into oracle/_ref/libdbg_ref.so (outputs only; no source
# is copied; no stand-in headers: without Boost this script stops with exit code 2 = "cannot pin here"), then runs its get_consensus and
# clairs_to_amd/realign/debruijn_graph.so (the product, same C ABI) on the generated windows of tests/consensus_enum.py (the 250 windows of
# the test-suite plus 2 000 more) and on the hand-derived vectors of tests/test_realign.py.
# Exit code = verdict: 0 every window equal (as sets AND in order), 1 a difference (printed), 2 not buildable here.
set -u
cd "$(dirname "$0")/.."
REFERENCE="${REFERENCE:-/root/reference}"
SRC="$REFERENCE/src/realign/debruijn_graph.cpp"
[ -f "$SRC" ] || { echo "pin_dbg: $SRC not found (set REFERENCE=)"; exit 2; }
INC=""
for d in "${BOOST_INC:-}" /usr/include /usr/local/include /opt/conda/include "${CONDA_PREFIX:-/nonexistent}/include"; do
    [ -n "$d" ] && [ -f "$d/boost/graph/adjacency_list.hpp" ] && { INC="$d"; break; }
done
[ -n "$INC" ] || { echo "pin_dbg: boost/graph/adjacency_list.hpp not found (set BOOST_INC=): cannot pin here"; exit 2; }
mkdir -p oracle/_ref
g++ -O2 -fPIC -w -std=c++14 -shared -I"$INC" -I"$REFERENCE/src/realign" "$SRC" -o oracle/_ref/libdbg_ref.so || { echo "pin_dbg: the reference did not compile"; exit 2; }
[ -f clairs_to_amd/realign/debruijn_graph.so ] || python -c "import __graft_entry__ as g; g.build()" || exit 2
exec python tools/pin_dbg.py oracle/_ref/libdbg_ref.so clairs_to_amd/realign/debruijn_graph.so
