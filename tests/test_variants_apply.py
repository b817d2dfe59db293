"""variants/*.patch are measured-and-rejected code kept as patches.  Each is anchored to a commit (variants/BASES); this
test checks that every patch has an entry and still applies to its base — through a scratch index, nothing is checked
out — so that scripts/build_patch_variant.sh can rebuild any of them.  Needs the repository's history (skipped in a
snapshot without .git, e.g. on the GPU box)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _git(*args, env=None):
    return subprocess.run(["git", "-C", ROOT] + list(args), capture_output=True, text=True, env=env)


@pytest.mark.skipif(not os.path.isdir(os.path.join(ROOT, ".git")), reason="no git history in this snapshot")
def test_every_variant_patch_applies_to_its_base(tmp_path):
    bases = {}
    for line in open(os.path.join(ROOT, "variants", "BASES")):
        if line.strip() and not line.startswith("#"):
            name, sha = line.split()[:2]
            bases[name] = sha
    patches = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "variants", "*.patch")))
    assert patches and set(patches) == set(bases), (sorted(set(patches) ^ set(bases)))
    readme = open(os.path.join(ROOT, "variants", "README.md")).read()
    for name in patches:
        assert "`%s`" % name in readme, name + " is not described in variants/README.md"
        if _git("cat-file", "-e", bases[name] + "^{commit}").returncode != 0:
            pytest.skip("shallow history: base commit %s not present" % bases[name][:8])
        env = dict(os.environ, GIT_INDEX_FILE=str(tmp_path / "idx"))
        assert _git("read-tree", bases[name], env=env).returncode == 0
        r = _git("apply", "--cached", "--check", os.path.join("variants", name), env=env)
        assert r.returncode == 0, (name, r.stderr[-400:])
