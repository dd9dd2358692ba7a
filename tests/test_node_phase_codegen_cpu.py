"""Round 6: properties of the SHIPPED code of k_chain16's node phase that its speed rests on and that only the compiler can break:
no flat loads (the step pointers come through the scalar cache, the LDS areas from an opaque LDS address -- not from look-ups of the dynamic-LDS base in
memory), fragment waits counted exactly (a handful of s_waitcnt vmcnt(0), not one per fragment group), 9 + 2 workgroup barriers.  No GPU needed."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("PS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
NODE = "_ZN2ps14c16_node_phaseILi8EEEvPKNS_9ChainStepES3_jiiifPy"


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="no llvm-objdump in this image")
def test_node_phase_of_the_shipped_library(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", ge.LIB, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f"--disassemble-symbols={NODE}", co], check=True, capture_output=True, text=True).stdout
    body = [l for l in dis.splitlines() if re.match(r"^\s+[sv]_|^\s+(ds|global|flat|buffer|scratch)_", l)]
    assert len(body) > 2000, "the node phase is in the library under its own symbol (a __noinline__ function)"
    count = lambda pat: sum(1 for l in body if re.search(pat, l))
    assert count(r"\bflat_") == 0                      # (a flat load = a pointer or an LDS base fetched from memory in the middle of the stages)
    assert count(r"\bscratch_") == 0                   # (no spills: the fragment ring lives in registers)
    assert count(r"v_mfma_f32_16x16x32_f16") >= 183    # (straight-line GEMMs: the 183 matrix instructions of a layer, every fragment group unrolled)
    assert count(r"s_waitcnt vmcnt\(0\)") <= 16        # (was 48 with run-time group counts: one full drain per fragment group)
    assert count(r"s_waitcnt vmcnt\((1[0-9]|2[0-9])\)") >= 20   # (exact waits: this group is in, two more stay in flight)
    assert count(r"s_barrier") == 11                   # 7 (POST) + 2 (PRE) + 2 (the launch's first PRE)


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="no llvm-objdump in this image")
def test_half_register_writes_of_the_inline_assembly_are_followed_by_wait_states(tmp_path):
    """Round 6: v_fma_mixlo_f16 / v_fma_mixhi_f16 with an fp16 source (ps_chain16.h MixAsm: inline assembly, which hipcc's
    hazard recogniser cannot see into) write HALF a register; gfx950 wants wait states before the register is read, and an experiments build whose
    scheduler put a v_mfma right behind one returned wrong sums.  In the shipped library every run of them ends in s_nop 1."""
    import __graft_entry__ as ge
    ge.build()
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", ge.LIB, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout
    runs = pending = 0
    for line in dis.splitlines():
        m = re.match(r"^\s+([sv]_\w+|ds_\w+|global_\w+|flat_\w+|buffer_\w+|scratch_\w+)\s*(.*?)\s*//", line)
        if not m:
            continue
        op, args = m.group(1), m.group(2)
        mine = op in ("v_fma_mixlo_f16", "v_fma_mixhi_f16") and "op_sel_hi:[1,0,0]" in args   # (an fp16 first source: the form only the inline assembly uses)
        if mine:
            pending = True
        elif pending:
            assert op == "s_nop" and int(args.split()[0]) >= 1, f"{op} {args} right behind a half-register write of the inline assembly"
            pending = False
            runs += 1
    assert runs >= 20 and not pending   # (the edge bodies of k_chain16, k_edge16, k_edge_rows, k_attn_chain<1, ..., GEO>: several groups each)
