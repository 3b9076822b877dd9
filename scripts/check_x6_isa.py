#!/usr/bin/env python3
"""Static check of the split-bf16 GEMM's main loop (fuxictr_amd/csrc/fx_gemm_x6.hip): compile to ISA and report,
for every kernel instance, the s_waitcnt vmcnt(N) of its plain k loop and how many instructions after the last
global_load each one sits.  A wait with N smaller than the loads just issued, right behind them, drains the
prefetch (a full memory latency per iteration) — the register allocator does that when it decides to shuffle
a freshly loaded tuple (round 5: production kernel 55 us vs the lab's 49 us on 4096 x 1024 x 1024)."""
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "fuxictr_amd/csrc/fx_gemm_x6.hip"
extra = sys.argv[2:]
asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950",
                      "-x", "hip", "-S", src, "-o", "-", "--cuda-device-only"] + extra, capture_output=True, text=True)
if asm.returncode:
    sys.exit(asm.stderr[-2000:])
cur, kernels = None, {}
for line in asm.stdout.splitlines():
    m = re.match(r"^(_Z\S+):", line)
    if m:
        cur = m.group(1)
        kernels[cur] = []
    elif cur and line.startswith("\t") and not line.strip().startswith((";", ".")):
        kernels[cur].append(line.strip())
    if "s_endpgm" in line:
        cur = None
for name, ins in kernels.items():
    if "gemm_x6" not in name:
        continue
    # the plain loop: the first stretch with 48 MFMAs between two backward branches is good enough a proxy:
    # report every vmcnt wait in the kernel with its distance to the previous global_load and the N
    out, last_load, nload_since = [], None, 0
    bad = 0
    for i, x in enumerate(ins):
        if x.startswith("global_load"):
            last_load = i
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", x)
        if m and last_load is not None:
            d = i - last_load
            n = int(m.group(1))
            tag = ""
            if d <= 6 and n <= 1:
                tag, bad = "  <-- drains the loads just issued", bad + 1
            out.append("vmcnt(%d)@+%d%s" % (n, d, tag))
    vg = [x for x in asm.stdout.splitlines() if ".vgpr_count" in x]
    print("%s: %d instrs, %d MFMA, %d v_mov, drains: %d" % (name[:40], len(ins), sum(x.startswith("v_mfma") for x in ins),
                                                            sum(x.startswith("v_mov") for x in ins), bad))
    print("   ", " ".join(out[:40]))
