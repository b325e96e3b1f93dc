"""Registers and scratch of the proving path's kernels, from the compiler's own metadata (no GPU: hipcc cross-compiles gfx950).
What runs beside what on a SIMD is decided by these numbers (csrc/kernels_msm.hpp "built to FIT"): two accumulation wavefronts hold
2 x 176 (G1) / 2 x 248 (G2) of a SIMD's 512 VGPRs; the G1 oversized-bucket / reduction kernels must fit in the 160 left beside two G1
accumulation wavefronts, the G2 ones in the 336 left beside one.  VERDICT r5 next 3: no kernel of zkr_prove / zkr_key uses scratch."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _kernels(unit, tmp_path):
    out = tmp_path / (unit + ".s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value",
                           "--cuda-device-only", "-S", os.path.join(ROOT, "simple-zk-rollups_amd", "csrc", unit + ".hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    txt = out.read_text()
    rows = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        g = lambda key: int(re.search(r"\.amdhsa_" + key + r"\s+(\d+)", m.group(2)).group(1))
        rows[m.group(1)] = (g("private_segment_fixed_size"), g("next_free_vgpr"))
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
    return {re.sub(r"\(.*", "", n).replace("zkr::", "").replace("void ", "").replace("Fp<FqParams>", "Fq"): v for n, v in zip(names, rows.values())}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_scratch_and_the_register_footprints_that_decide_co_residency(tmp_path):
    ks = _kernels("zkr_prove", tmp_path)
    assert len(ks) >= 40
    assert {k: v for k, v in ks.items() if v[0]} == {}                      # no kernel of the proving path spills or keeps a stack frame
    alloc = lambda n: (n + 7) // 8 * 8                                      # VGPRs are handed out in granules of 8
    g1_acc = alloc(ks["msm_accum_kernel<Fq, 2, true>"][1])
    g2_acc = alloc(ks["msm_accum_kernel<Fq2, 2, false>"][1])
    assert 2 * g1_acc <= 512 < 3 * g1_acc and 2 * g2_acc <= 512             # two accumulation wavefronts per SIMD, and not three of G1
    for name in ("msm_big_kernel", "msm_big_finish_kernel", "msm_reduce1_kernel", "msm_reduce2_kernel", "msm_reduce3_kernel"):
        assert 2 * g1_acc + alloc(ks[name + "<Fq >"][1]) <= 512, name       # a third wavefront beside two G1 accumulation wavefronts
        assert g1_acc + alloc(ks[name + "<Fq2>"][1]) <= 512, name            # beside one
    assert 2 * g1_acc + alloc(max(v[1] for k, v in ks.items() if k.startswith("ntt_pass_kernel"))) <= 512   # an NTT wavefront as well


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_key_build_kernels_use_no_scratch(tmp_path):
    assert {k: v for k, v in _kernels("zkr_key", tmp_path).items() if v[0]} == {}
