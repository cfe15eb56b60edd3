"""registers / scratch / spills of every kernel of one csrc/*.hip file (cross-compiles, no GPU): python tools/kernel_resources.py rollout_chain.hip [-D...]"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "constrainedcontrol.jl_amd", "csrc", sys.argv[1])
asm = "/tmp/" + os.path.basename(src) + ".s"
t0 = time.time()
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=fast", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", asm, src] + sys.argv[2:],
                      stderr=subprocess.DEVNULL)
txt = open(asm).read()
print("compiled in %.0f s -> %s" % (time.time() - t0, asm))
for blk in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
    print("%-75s agpr %3d vgpr %3d scratch %4d sgpr_spill %3d vgpr_spill %3d" % (name[:75], int(blk.split()[0]), g("vgpr_count"), g("private_segment_fixed_size"), g("sgpr_spill_count"), g("vgpr_spill_count")))
