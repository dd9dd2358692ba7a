"""bench.py's one-line JSON contract, checked without a GPU: the keys the driver reads must be spelled in the
dict literal that rank 0 prints (a comment once swallowed one)."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
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
