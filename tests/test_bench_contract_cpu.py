"""bench.py's one-line JSON contract, checked without a GPU: the keys the driver reads must be spelled in the
dict literal that rank 0 prints (a comment once swallowed one)."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "traffic_over_algorithmic_bytes", "traffic_source"}
CPU_BASELINE = {"value", "unit", "cores", "kind", "sample"}


def _dict_keys(node):
    return {k.value for k in node.keys if isinstance(k, ast.Constant)}


def test_bench_json_line_has_the_contract_keys():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    dicts = [n for n in ast.walk(tree) if isinstance(n, ast.Dict)]
    top = [d for d in dicts if REQUIRED <= _dict_keys(d)]
    assert top, "no dict literal in bench.py carries all contract keys"
    roof = [d for d in dicts if ROOFLINE <= _dict_keys(d)]
    assert roof, "roofline object incomplete"
    calls = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "dict"]
    assert any(CPU_BASELINE <= {k.arg for k in c.keywords} for c in calls), "cpu_baseline object incomplete"


def test_algorithmic_bytes_of_the_policy_launch():
    """roofline.algorithmic_bytes (round 6): weights + k | v rows + geometry records + rows in / out of the default workload's policy launch --
    the figure the counter bytes (roofline.traffic) are read against."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    p = b.algorithmic_bytes_chain(1024, 1024, 8192, 63112, 164283, 6)
    assert p["weights"] == 12 * 960 * 1024 and p["kv_rows_m2p"] == 6 * 8192 * 1024 and p["geometry_records"] == (63112 + 164283) * 32
    assert p["total"] == sum(v for k, v in p.items() if k != "total") and 70e6 < p["total"] < 80e6
    f = b.newest_pmc_json()
    assert f and os.path.basename(f).endswith("_pmc_policy_chain.json")
    aff = b.pin_to_gpu_numa_node(0)   # (no GPU here: says why it did nothing, never raises)
    assert aff["pinned"] is False and "why" in aff
