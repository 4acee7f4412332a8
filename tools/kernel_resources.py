#!/usr/bin/env python3
"""Registers, LDS, scratch and the resulting waves per SIMD of every kernel in a built library (no GPU needed):
   python tools/kernel_resources.py [mpc-ilqr-mujoco_amd/lib/libilqr_hip.so]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "mpc-ilqr-mujoco_amd", "lib", "libilqr_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    tmp = os.path.join(d, os.path.basename(lib))
    os.symlink(lib, tmp)
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", tmp], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    for co in sorted(glob.glob(os.path.join(d, "*amdgcn*"))):
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "0"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip().split("(")[0].replace("ilqr::", "").replace("void ", "")
            agpr = int(re.match(r"\s*(\d+)", blk).group(1))
            vg, lds, scr = int(g("vgpr_count")), int(g("group_segment_fixed_size")), int(g("private_segment_fixed_size"))
            tot = vg  # on gfx90a+ .vgpr_count is the unified (arch + acc) allocation
            waves = max(1, min(8, 512 // max(1, ((tot + 7) // 8) * 8)))
            rows.append((name, vg, agpr, int(g("sgpr_count")), lds, scr, int(g("vgpr_spill_count")), waves, int(g("max_flat_workgroup_size"))))
    print("%-34s %5s %5s %5s %8s %8s %6s %10s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "lds B", "scratch", "spill", "waves/SIMD", "wg"))
    for r in sorted(rows):
        print("%-34s %5d %5d %5d %8d %8d %6d %10d %6d" % r)
