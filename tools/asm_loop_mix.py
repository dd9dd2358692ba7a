"""Static instruction mix of one function (and of its innermost hot loop) in hipcc's gfx950 assembly -- no GPU needed.
usage: python tools/asm_loop_mix.py <file.s> <symbol substring> [loop label]
With no loop label: the function's totals, its resources, and every Depth=2 loop header found (the tile loop of the edge
phase is the one that contains the v_sin_f32 block).  The slow-path child loops (libm sincosf) are excluded from a loop's count."""
import collections, re, sys

def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op in ("v_sin_f32_e32", "v_cos_f32_e32", "v_exp_f32_e32", "v_rcp_f32_e32", "v_rsq_f32_e32", "v_log_f32_e32", "v_sqrt_f32_e32",
              "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_rcp_f32"): return "trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    path, sym = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().endswith(":") or (l.startswith("_Z") and sym in l.split(":")[0] and ":" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    print("function:", lines[start].split(":")[0][:120], "lines", start + 1, "-", end + 1)
    for l in lines[end:end + 60]:
        if any(k in l for k in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "NumSgprs")): print("  ", l.strip("; \t"))
    heads = [(i, l) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    if label is None:
        for i, l in heads:
            ctx = " ".join(x.strip() for x in body[i:i + 4])
            if "Depth=2" in ctx and "Loop Header" in ctx: print("  depth-2 loop header:", l.split(":")[0])
        return count(body, "whole function")
    # blocks of the loop: the header block + every block whose label comment says "in Loop: Header=<label> Depth=2" (child loops carry
    # their own header's name and are left out)
    key = "Header=" + label.lstrip(".L") + " "
    seg, take = [], False
    for i, l in enumerate(body):
        if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
            ctx = l + " " + " ".join(x for x in body[i + 1:i + 4] if x.strip().startswith(";"))
            take = l.startswith(label + ":") or key in ctx
        if take: seg.append(l)
    count(seg, "loop " + label + " (child loops excluded)")

def count(seg, title):
    c = collections.Counter(); ops = collections.Counter()
    for l in seg:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") : continue
        op = t.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op): continue
        c[classify(op)] += 1; ops[op] += 1
    tot_valu = c["valu"] + c["valu_pk"] + c["trans"]
    print(f"{title}: VALU+trans {tot_valu} (plain {c['valu']}, packed {c['valu_pk']}, trans {c['trans']}), MFMA {c['mfma']}, LDS {c['lds']}, VMEM {c['vmem']}, SALU {c['salu']}, s_nop {c['s_nop']}, s_waitcnt {c['s_waitcnt']}")
    print("  top VALU ops:", ", ".join(f"{k} {v}" for k, v in ops.most_common(60) if k.startswith("v_") and not k.startswith("v_mfma"))[:1500])

main()
