"""Build step of libprosim_hip.so (called by __graft_entry__.build): rewrites the device assembly hipcc produced so that no packed-fp32
VALU instruction carries an op_sel bit.

Why (tools/mb/mb_pksgpr3.hip reproduces it in 3 s on an MI355X): on gfx950 a v_pk_fma_f32 / v_pk_mul_f32 /
v_pk_add_f32 whose op_sel routes the HIGH register of a source pair into the LOW half of the result returns a wrong low half in lanes
48-63 -- now and then, and only while a wave of ANOTHER kernel on the same SIMD issues v_mfma_f32_16x16x32_f16.  Alone on the GPU the
instruction is exact, which is why every single-engine parity test passed while 2 - 13 % of the encoder runs beside other engines
differed (k_edge_geo's LayerNorm statistics: `v_pk_fma_f32 v[60:61], v[58:59], s[12:13], v[18:19] op_sel:[0,0,1]`).  hipcc emits such
forms on its own (SLP vectorisation, broadcasts of a pair's second element); the hazard is not in its tables.

The pass replaces every such instruction by the two 32-bit instructions it stands for (same operands, same rounding: the results are
bit-identical), low half first unless that would overwrite a register the high half still reads.  Packed instructions without op_sel
(natural lane order, op_sel_hi broadcasts of the FIRST register) were never seen to fail and are left alone.

usage: python pk_legalize.py in.s out.s   (prints how many instructions it split, per kernel with -v)
       python pk_legalize.py --check file.s   (exit 1 if a packed-fp32 instruction with op_sel is left)"""
from __future__ import annotations

import re
import sys

PK = re.compile(r"^(\s*)v_pk_(fma|mul|add)_f32\s+(.*)$")
MOD = re.compile(r"\b(op_sel_hi|op_sel|neg_lo|neg_hi):\[([01,]+)\]")


def _elem(op: str, hi: int) -> str:
    """Register (or constant) holding element `hi` of a packed source."""
    m = re.match(r"^([vs])\[(\d+):(\d+)\]$", op)
    if m:
        return "%s%d" % (m.group(1), int(m.group(2)) + hi)
    if op in ("vcc", "exec"):
        return "%s_%s" % (op, "hi" if hi else "lo")
    if re.match(r"^-?(\d+(\.\d+)?(e[-+]?\d+)?|0x[0-9a-fA-F]+)$", op):   # inline constant: the value sits in the low element; the high one is 0
        if hi:
            raise ValueError("op_sel selects the high element of a constant: " + op)
        return op
    raise ValueError("unknown packed operand: " + op)


def split(line: str):
    """None if the line is not a packed-fp32 instruction with an op_sel bit; else the replacement lines."""
    m = PK.match(line.split(";")[0].rstrip("\n"))
    if not m:
        return None
    indent, opc, rest = m.groups()
    mods = {k: [int(x) for x in v.split(",")] for k, v in MOD.findall(rest)}
    if not any(mods.get("op_sel", [])):
        return None
    clamp = bool(re.search(r"\bclamp\b", rest))
    ops = [o.strip() for o in MOD.sub("", re.sub(r"\bclamp\b", "", rest)).strip().rstrip(",").split(",")]
    ops = [o for o in ops if o]
    dst, srcs = ops[0], ops[1:]
    n = len(srcs)
    assert n == (3 if opc == "fma" else 2), line
    sel = mods.get("op_sel", [0] * n)
    sel_hi = mods.get("op_sel_hi", [1] * n)
    neg_lo = mods.get("neg_lo", [0] * n)
    neg_hi = mods.get("neg_hi", [0] * n)
    d = re.match(r"^v\[(\d+):(\d+)\]$", dst)
    assert d, line
    dlo, dhi = "v%d" % int(d.group(1)), "v%d" % (int(d.group(1)) + 1)
    lo_src = [_elem(s, sel[j]) for j, s in enumerate(srcs)]
    hi_src = [_elem(s, sel_hi[j]) for j, s in enumerate(srcs)]

    def emit(dreg, regs, negs):
        args = ", ".join(("-" if negs[j] else "") + r for j, r in enumerate(regs))
        mnem = {"fma": "v_fma_f32", "mul": "v_mul_f32_e64", "add": "v_add_f32_e64"}[opc]
        return "%s%s %s, %s%s" % (indent, mnem, dreg, args, " clamp" if clamp else "")

    lo_i, hi_i = emit(dlo, lo_src, neg_lo), emit(dhi, hi_src, neg_hi)
    tag = "%s; (pk_legalize: was v_pk_%s_f32 %s)" % (indent, opc, rest.strip())
    if dlo not in hi_src:
        return [tag, lo_i, hi_i]
    if dhi not in lo_src:
        return [tag, hi_i, lo_i]
    # both halves read both destination registers.  The horizontal forms (v_pk_add_f32 v[2:3], v[2:3], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]:
    # v2 + v3 in both halves) compute ONE value: a + b == b + a and a * b == b * a bit for bit, so the second half is a copy
    same = (lo_src == hi_src and neg_lo == neg_hi) if opc == "fma" else (sorted(zip(lo_src, neg_lo)) == sorted(zip(hi_src, neg_hi)))
    if opc == "fma" and not same:   # (fma: the two multiplicands commute, the addend does not)
        same = sorted(zip(lo_src[:2], neg_lo[:2])) == sorted(zip(hi_src[:2], neg_hi[:2])) and (lo_src[2], neg_lo[2]) == (hi_src[2], neg_hi[2])
    if same:
        return [tag, lo_i, "%sv_mov_b32_e32 %s, %s" % (indent, dhi, dlo)]
    # a crossed pair (v_pk_mul_f32 v[10:11], v[8:9], v[10:11] op_sel:[0,1] op_sel_hi:[0,0]: low = v8 * v11, high = v8 * v10): exchange the two
    # destination registers first, then every half reads its own
    sw = {dlo: dhi, dhi: dlo}
    lo_sw, hi_sw = [sw.get(r, r) for r in lo_src], [sw.get(r, r) for r in hi_src]
    if dlo not in hi_sw:
        return [tag, "%sv_swap_b32 %s, %s" % (indent, dlo, dhi), emit(dlo, lo_sw, neg_lo), emit(dhi, hi_sw, neg_hi)]
    if dhi not in lo_sw:
        return [tag, "%sv_swap_b32 %s, %s" % (indent, dlo, dhi), emit(dhi, hi_sw, neg_hi), emit(dlo, lo_sw, neg_lo)]
    raise ValueError("both orders overwrite a source (needs a temporary): " + line.strip())


# PS_ASM_NOP_AFTER_SWAP=1 (tools only): two wait states behind every v_permlane16/32_swap_b32 -- does a consumer right behind the swap matter?
_NOP_AFTER_SWAP = bool(int(__import__("os").environ.get("PS_ASM_NOP_AFTER_SWAP", "0")))


def legalize(text: str):
    out, counts, name = [], {}, None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".") and not line.startswith("\t"):
            name = m.group(1)
        rep = split(line)
        if rep is None:
            out.append(line)
            if _NOP_AFTER_SWAP and re.match(r"^\s*v_permlane(16|32)_swap_b32", line):   # (experiment switch: see the module's end)
                out.append("\ts_nop 1")
        else:
            out.extend(rep)
            counts[name] = counts.get(name, 0) + 1
    return "\n".join(out) + "\n", counts


def remaining(text: str):
    """Packed-fp32 instructions that still carry an op_sel bit (assembly text or llvm-objdump output)."""
    bad = []
    for line in text.splitlines():
        body = re.sub(r"^\s*[0-9a-f]+:\s+", "", line.split("//")[0].split(";")[0]).strip()   # (objdump prefixes an address)
        m = re.match(r"^v_pk_(fma|mul|add)_f32\s+(.*)$", body)
        if m and any(int(x) for k, v in MOD.findall(m.group(2)) if k == "op_sel" for x in v.split(",")):
            bad.append(body)
    return bad


def main(argv):
    if len(argv) >= 3 and argv[1] == "--check":
        bad = remaining(open(argv[2]).read())
        for b in bad[:20]:
            print("  " + b)
        print("packed-fp32 instructions with an op_sel bit: %d" % len(bad))
        return 1 if bad else 0
    src, dst = argv[1], argv[2]
    text, counts = legalize(open(src).read())
    open(dst, "w").write(text)
    if "-v" in argv:
        for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
            print("%5d  %s" % (v, k))
    print("pk_legalize: %d packed-fp32 instructions with op_sel split into 32-bit pairs (%d kernels)" % (sum(counts.values()), len(counts)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
