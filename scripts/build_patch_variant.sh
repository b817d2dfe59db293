#!/bin/bash
# Builds one of variants/*.patch:  scripts/build_patch_variant.sh <patch name without .patch>  ->  rpt_amd/lib/librptgpu_<name>.so
# The patches are anchored to commits (variants/BASES): the base commit is checked out into a scratch worktree, the patch
# applied there — or the script FAILS, loudly — and the library built with that tree's own Makefile.
set -euo pipefail
NAME=$1
HERE=$(cd "$(dirname "$0")/.." && pwd)
BASE=$(awk -v p="$NAME.patch" '$1 == p {print $2}' "$HERE/variants/BASES")
[ -n "$BASE" ] || { echo "build_patch_variant: variants/BASES has no entry for $NAME.patch" >&2; exit 2; }
W=$(mktemp -d /tmp/rpt_variant_XXXX)
trap 'git -C "$HERE" worktree remove --force "$W" >/dev/null 2>&1 || true' EXIT
git -C "$HERE" worktree add --detach "$W" "$BASE" >/dev/null
git -C "$W" apply --check "$HERE/variants/$NAME.patch" || { echo "build_patch_variant: $NAME.patch does NOT apply to its base $BASE" >&2; exit 3; }
git -C "$W" apply "$HERE/variants/$NAME.patch"
make -s -j8 -C "$W/rpt_amd/csrc"
cp "$W/rpt_amd/lib/librptgpu.so" "$HERE/rpt_amd/lib/librptgpu_$NAME.so"
echo "built rpt_amd/lib/librptgpu_$NAME.so from $BASE + $NAME.patch (use with RPTGPU_LIB; its ABI is that commit's)"
