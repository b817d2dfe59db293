#!/usr/bin/env python3
"""Print VGPR / AGPR / SGPR counts, spills, scratch and static LDS of every kernel in a HIP object or code object:
   python scripts/kernel_resources.py rpt_amd/csrc/build/kernels_strict.o [name-filter]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
def notes(path):
    with tempfile.TemporaryDirectory() as d:
        co = path
        if not path.endswith(".co"):
            import shutil
            tmp = os.path.join(d, "in.o"); shutil.copy(path, tmp)
            subprocess.run([LLVM + "/llvm-objdump", "--offloading", tmp], cwd=d, check=True, stdout=subprocess.DEVNULL)
            co = [os.path.join(d, f) for f in os.listdir(d) if "amdgcn" in f][0]
        return subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
def main():
    t = notes(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", t, re.S):
        b = m.group(0); name = re.search(r"\.name:\s+(\S+)", b).group(1)
        if flt not in name or (not flt and "rocprim" in name): continue
        g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print("%-60s vgpr %3s agpr %3s sgpr %3s vspill %3s sspill %3s scratch %5s lds %5s" % (
            dem[:60], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
            g("private_segment_fixed_size"), g("group_segment_fixed_size")))
main()
