#!/usr/bin/env python3
"""Registers and scratch of every kernel of a translation unit, from the compiler's own metadata (no GPU needed):
   python tools/kernel_resources.py zkr_prove [zkr_key ...]      (names of csrc/*.hip; --all-kernels also lists the scratch-free ones)
compiles csrc/<name>.hip to gfx950 assembly (device side only) and prints, per kernel: scratch bytes per lane, VGPRs, AGPRs."""
import os, re, subprocess, sys, tempfile
CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "simple-zk-rollups_amd", "csrc")
show_all = "--all-kernels" in sys.argv
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
for name in [a for a in sys.argv[1:] if not a.startswith("-")]:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, name + ".s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value",
                               "--cuda-device-only", "-S", os.path.join(CSRC, name + ".hip"), "-o", out] + extra, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        body = m.group(2)
        g = lambda key: int(re.search(r"\.amdhsa_" + key + r"\s+(\d+)", body).group(1))
        rows.append((m.group(1), g("private_segment_fixed_size"), g("next_free_vgpr"), g("accum_offset")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("== %s.hip: %d kernels, %d with scratch" % (name, len(rows), sum(1 for r in rows if r[1])))
    for (mangled, scratch, vgpr, acc), nm in zip(rows, names):
        if scratch or show_all:
            nm = re.sub(r"\(.*", "", nm).replace("zkr::", "").replace("void ", "").replace("Fp<FqParams>", "Fq")
            print("  scratch %4d B  vgpr+agpr %3d (arch %3d)  %s" % (scratch, vgpr, acc, nm))
