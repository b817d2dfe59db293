"""The product path must not import, link or execute anything under oracle/."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_sources_do_not_reference_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "rpt_amd")):
        if "build" in base or "__pycache__" in base:
            continue
        for f in files:
            if not f.endswith((".py", ".cpp", ".h", ".hip", ".inc", "Makefile")):
                continue
            txt = open(os.path.join(base, f), errors="ignore").read()
            if re.search(r"oracle", txt, flags=re.I):
                bad.append(os.path.join(base, f))
    assert not bad, bad


def test_product_library_does_not_link_the_oracle():
    lib = os.path.join(ROOT, "rpt_amd", "lib", "librptgpu.so")
    out = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
    assert "oracle" not in out
    assert "amdhip64" in out
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    assert "oracle_" not in syms
