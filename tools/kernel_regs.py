"""Register / LDS use of every kernel: python tools/kernel_regs.py  (compiles fw_kernels.hip to assembly under /tmp)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/fw_kernels_regs.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950", "-x", "hip",
                       "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "firework_amd/csrc/fw_kernels.hip")] + sys.argv[1:], stderr=subprocess.DEVNULL)
t = open(out).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", t):
    name = re.sub(r"^_ZN2fw\d+", "", m.group(1))
    name = re.sub(r"(ENS_|EN2fw|IL[bj]).*", "", name)
    v = int(m.group(3)); alloc = (v + 7) // 8 * 8
    print(f"{m.group(1)[:70]:70s} sgpr {m.group(2):>3s} vgpr {v:3d} -> {min(8, 512 // alloc)} waves/SIMD  spills {m.group(4)}")
