"""CPU test (no GPU): pins two properties of the compiled weight-gradient kernel that its speed depends on and that a
compiler bump could silently break ("harden the build").  The kernel issues the NEXT tile's LDS-DMA transfers
(`buffer_load_dwordx4 ... lds`, inline assembly: csrc/common.hpp lds_dma16_async) right in front of the MFMA phase of the
CURRENT tile; hipcc used to put `s_waitcnt vmcnt(0)` between the two (it models an LDS-DMA as a store to all of LDS), which
serialised transfer and compute (-7 %, DESIGN.md section 3 "Round 2").

  1. every LDS-DMA instruction is directly preceded by ITS `s_mov_b32 m0, <sgpr>` (+ the `s_nop 0` the hazard rule wants):
     M0 is compiler-reserved, the statement saves / writes / restores it itself;
  2. between the last LDS-DMA of a batch and the first transposing LDS read (`ds_read_b64_tr_b16`) of the MFMA phase that
     follows it there is no `s_waitcnt vmcnt(0)`.

The device code is taken from the in-tree object file (built by __graft_entry__.build(); built here if missing)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "brats21_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
KERNEL = "_Z26conv_wgrad_alltaps2_kernelILi3ELi3EEv11WgradParams"


def _disassemble(name="conv_wgrad.o"):
    obj = os.path.join(CSRC, name)
    if not os.path.exists(obj):
        subprocess.run(["make", "-C", CSRC, name], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tmp = tempfile.mkdtemp(prefix="brats_isa_")
    try:
        shutil.copy(obj, os.path.join(tmp, "k.o"))
        subprocess.run([OBJDUMP, "--offloading", "k.o"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f]
        assert dev, f"no gfx950 code object in {name}"
        return subprocess.run([OBJDUMP, "-d", dev[0]], cwd=tmp, check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _kernel_instructions(text, name):
    lines = text.splitlines()
    start = next(i for i, l in enumerate(lines) if l.rstrip().endswith(f"<{name}>:"))
    out = []
    for l in lines[start + 1:]:
        if re.match(r"^[0-9a-f]+ <", l):  # next symbol
            break
        ins = l.split("//")[0].strip()
        if ins:
            out.append(ins)
    return out


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_wgrad_lds_dma_is_not_waited_for_before_the_mfma_phase():
    ins = _kernel_instructions(_disassemble(), KERNEL)
    dma = [i for i, s in enumerate(ins) if s.startswith("buffer_load_dwordx4") and s.endswith("lds")]
    assert len(dma) >= 16, f"expected two batches of 8 LDS-DMA instructions, found {len(dma)}"
    for i in dma:
        window = ins[max(0, i - 3):i]
        assert any(re.match(r"s_mov_b32 m0, s\d+", w) for w in window), (i, window)
    checked = 0
    for i in dma:
        for j in range(i + 1, len(ins)):
            if ins[j].startswith("ds_read_b64_tr_b16"):
                between = ins[i + 1:j]
                if not any(s.startswith("buffer_load_dwordx4") and s.endswith("lds") for s in between):  # i is the batch's last
                    bad = [s for s in between if s.startswith("s_waitcnt") and "vmcnt(0)" in s]
                    assert not bad, f"LDS-DMA at {i} is waited for before the MFMA phase: {bad}"
                    checked += 1
                break
            if ins[j].startswith("s_barrier") or ins[j].startswith("s_endpgm"):
                break  # (the first tile's batch runs into the loop-top wait: that one is meant)
    assert checked >= 1, "no LDS-DMA batch in front of an MFMA phase found: the kernel's structure changed"


def _classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_dominant_kernel_mma_loop_instruction_mix():
    """VERDICT r5 item 4 asked whether vector-instruction issue (the PMC's 1.98 VALU per MFMA over the whole kernel) is what holds
    the dominant kernel at 0.59 MFMA-busy.  The compiled code answers where those instructions are: the MMA region of
    conv_igemm_vs8_kernel<24, 1, 3> -- everything between the first and the last MFMA of a chunk body, 21 macro-steps x 24 MFMAs --
    carries 0.15 vector instructions per MFMA (address updates of the LDS reads and the weight stream), no scratch access, one
    LDS read per 3 MFMAs and one weight load per 8; the other ~1400 of the kernel's ~1500 vector instructions are the per-chunk
    staging and the per-tile epilogue (statistics, 16-bit packing, stores).  Pinned here so that a compiler bump that drags address
    arithmetic into the loop, or spills in it, is noticed; DESIGN.md (dominant-kernel table, round 6) draws the conclusion."""
    ins = [i.split()[0] for i in _kernel_instructions(_disassemble("conv_bf16_k3_d1.o"), "_Z21conv_igemm_vs8_kernelILi24ELi1ELi3ELb0ELb0EEv10ConvParamsi")]
    mf = [i for i, op in enumerate(ins) if op.startswith("v_mfma")]
    assert len(mf) == 21 * 24, len(mf)  # one chunk body: ceil(27 taps x 3 units / 4) macro-steps x (8 voxel x 3 cout fragments)
    loop = [_classify(op) for op in ins[mf[0]:mf[-1] + 1]]
    n = {k: loop.count(k) for k in ("mfma", "valu", "lds", "vmem", "scratch")}
    print(f"\nconv_igemm_vs8<24,1,3> MMA region: {n}; whole kernel: {len(ins)} instructions, "
          f"{sum(1 for op in ins if _classify(op) == 'valu')} vector")
    assert n["scratch"] == 0
    assert n["valu"] <= 0.25 * n["mfma"], n
    assert n["lds"] <= 0.40 * n["mfma"] and n["vmem"] <= 0.20 * n["mfma"], n
    total_valu = sum(1 for op in ins if _classify(op) == "valu")
    assert total_valu - n["valu"] > 10 * n["valu"]  # the kernel's vector work sits outside the loop
