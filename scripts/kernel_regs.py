"""Register / spill table of every kernel of one .hip file (cross-compiled for gfx950, no GPU needed):
    python scripts/kernel_regs.py camera_calibrator_amd/csrc/cc_rig.hip [-DFLAG ...] [filter]
Reads the .amdhsa metadata of the generated assembly: vgpr / agpr / sgpr counts, spilled registers, scratch bytes."""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]
flags = [a for a in sys.argv[2:] if a.startswith("-")]
filt = [a for a in sys.argv[2:] if not a.startswith("-")]
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out, src] + flags,
                      stderr=subprocess.DEVNULL)
txt = open(out).read()
meta = txt[txt.index("amdhsa.kernels:"):]
rows = []
for blk in meta.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(cc::\w+(, .*)?\)$", "", name).replace("void cc::", "")
    rows.append((name, g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print("%-44s %5s %5s %5s %7s %7s %8s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
for r in rows:
    if not filt or any(f in r[0] for f in filt):
        print("%-44s %5s %5s %5s %7s %7s %8s %7s" % r)
