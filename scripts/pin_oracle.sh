#!/bin/bash
# Pins oracle/ to the REFERENCE on any machine that has cargo and network (this image has neither): clones rpt, applies
# rust/rpt.patch, renders the golden configurations on the CPU with the oracle's Philox stream behind rand's own
# distributions (feature `philox`), compares the dumped frames with the oracle and stores them as fixtures.
#
#   bash scripts/pin_oracle.sh [RPT_CHECKOUT]        # RPT_CHECKOUT: an existing clone of ekzhang/rpt (default: a fresh one)
#
# Writes tests/golden/ref_<scene>.npz (what scripts/compare_rust_golden.py --save-fixtures writes) and prints the
# comparison.  After that tests/test_golden.py holds the oracle to the reference's own output and the "parity
# unpinned" notes in oracle/oracle.cpp's header and DESIGN.md §3.1 can go.  The three commands are INTEGRATION.md §1's.
set -euo pipefail
HERE=$(cd "$(dirname "$0")/.." && pwd)
command -v cargo >/dev/null || { echo "pin_oracle: cargo not found — run this on a machine with a Rust toolchain" >&2; exit 2; }
WORK=$(mktemp -d)
SRC=${1:-}
if [ -z "$SRC" ]; then
  git clone --depth 1 https://github.com/ekzhang/rpt "$WORK/rpt"
  SRC=$WORK/rpt
fi
cp -r "$HERE/rust/rpt-gpu-sys" "$(dirname "$SRC")/" 2>/dev/null || true   # the patch's Cargo.toml points at ../rpt-gpu-sys
( cd "$SRC" && patch -p1 --forward < "$HERE/rust/rpt.patch" )
( cd "$SRC" && cargo run --release --features philox --example dump_golden -- "$WORK/golden" )
make -s -C "$HERE/oracle"
python "$HERE/scripts/compare_rust_golden.py" "$WORK/golden" --save-fixtures
echo "pin_oracle: fixtures written under $HERE/tests/golden/ (ref_*.npz); commit them"
