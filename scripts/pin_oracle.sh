#!/bin/bash
# ONE command that pins oracle/ — and through tests/test_rust_golden.py the HIP path — to the REFERENCE, on any machine
# that has cargo (this image has neither cargo nor network):
#
#   bash scripts/pin_oracle.sh [RPT_CHECKOUT]     # RPT_CHECKOUT: an existing clone of ekzhang/rpt (default: a fresh one)
#
# It copies the checkout, applies rust/rpt.patch (the `philox` feature: the oracle's Philox stream behind rand's own
# distributions; `Renderer::seed`; the dump_golden example), renders the golden configurations with the Rust program on
# the CPU, renders the same configurations with the oracle, compares, writes tests/golden/rust_<scene>.npz and ends with
#     pin_oracle: PASS ...        (exit 0)   or   pin_oracle: FAIL (...)   (exit 1; exit 2: no cargo)
# After a PASS: commit tests/golden/rust_*.npz; `pytest tests/test_rust_golden.py` (CPU: oracle == Rust; -m gpu: HIP ==
# oracle == Rust) stops skipping, and the "parity unpinned" notes in oracle/oracle.cpp's header and DESIGN.md can go.
# CARGO_FLAGS (e.g. --offline for a vendored registry) is passed to cargo.
set -uo pipefail
HERE=$(cd "$(dirname "$0")/.." && pwd)
fail() { echo "pin_oracle: FAIL ($1)"; exit "${2:-1}"; }
command -v cargo >/dev/null || fail "cargo not found — run this on a machine with a Rust toolchain" 2
WORK=$(mktemp -d)
SRC=${1:-}
if [ -z "$SRC" ]; then
  git clone --depth 1 https://github.com/ekzhang/rpt "$WORK/upstream" || fail "git clone of ekzhang/rpt failed (no network? pass a checkout)"
  SRC=$WORK/upstream
fi
[ -f "$SRC/Cargo.toml" ] && [ -f "$SRC/src/renderer.rs" ] || fail "$SRC is not a checkout of ekzhang/rpt"
# work on a COPY: the caller's checkout stays as it is, and a second run starts from clean sources again
mkdir -p "$WORK/rpt" && cp -r "$SRC/." "$WORK/rpt/" || fail "copying the checkout failed"
cp -r "$HERE/rust/rpt-gpu-sys" "$WORK/rpt-gpu-sys"            # the patch's Cargo.toml points at ../rpt-gpu-sys (feature `gpu`, unused here)
( cd "$WORK/rpt" && patch -p1 --forward < "$HERE/rust/rpt.patch" ) || fail "rust/rpt.patch does not apply to $SRC"
( cd "$WORK/rpt" && cargo run ${CARGO_FLAGS:-} --release --features philox --example dump_golden -- "$WORK/golden" ) \
  || fail "cargo run --features philox --example dump_golden failed"
make -s -C "$HERE/oracle" || fail "building the oracle failed"
python "$HERE/scripts/compare_rust_golden.py" "$WORK/golden" --rpt-root "$WORK/rpt" --save-fixtures
rc=$?   # (compare_rust_golden.py printed the PASS / FAIL line)
[ $rc -eq 0 ] && echo "next: git add tests/golden/rust_*.npz && python -m pytest tests/test_rust_golden.py"
exit $rc
