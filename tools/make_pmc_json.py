"""gpurun_out/<tag>_pmc_chain16.txt (tools/gpu_pmc_chain16.sh) -> profiles/<round>_pmc_policy_chain.json (round = the tag up to its first underscore): the HBM-side bytes of
one k_chain16 policy launch behind bench.py's roofline.traffic, stamped with its source.
usage: python tools/make_pmc_json.py <tag> <rows> <git hash>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rows, git = sys.argv[1], int(sys.argv[2]), sys.argv[3]
vals = {}
for line in open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_chain16.txt")):
    m = re.match(r"(\S+)\s+launches\s+(\d+)\s+avg per launch\s+([0-9.eE+-]+)", line)
    if m:
        vals[m.group(1)] = (int(m.group(2)), float(m.group(3)))
fetch_kb, write_kb = vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]
out = {
    "kernel": f"k_chain16<8, true, true> = the policy launch of bench.py's default (throughput) mode ({rows} rows per 8-wave workgroup, "
              f"12 layers, 1024 agents = 8 x configs[2] scenes), alone on the GPU",
    "source": f"profiles/{tag}_pmc_chain16.txt: rocprofv3 --kernel-trace --pmc <group>, one group per pass (tools/gpu_pmc_chain16.sh {tag} {rows} "
              f"over tools/gpu_c16_prof.py); FETCH_SIZE and WRITE_SIZE in passes of their own",
    "git": git, "chain_rows": rows, "launches_averaged": vals["FETCH_SIZE"][0],
    "fetch_size_kb_avg": fetch_kb, "write_size_kb_avg": write_kb,
    "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
    "note": "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (calibrated in round 1: profiles/r01_g_pmc_calibration.txt, "
            "a 1 GiB 16 B/lane streaming read reports exactly half) + WRITE_SIZE; Infinity-Cache hits are counted.  Round 1's launch "
            "(k_attn_chain, rel-PE operand images) moved 1.33 GB, round 2's (k_chain16 with the phase exchange and callee-saved registers "
            "through global scratch) 0.28 GB of which 77 MB were writes.",
    "counters": {k: v[1] for k, v in vals.items()},
}
rnd = tag.split("_")[0]
with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_policy_chain.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: out[k] for k in ("chain_rows", "fetch_size_kb_avg", "write_size_kb_avg", "hbm_bytes_per_launch")}))
