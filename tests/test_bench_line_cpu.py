"""The driver parses ONE JSON line from bench.py's stdout; round 4's grew to 31 KB and came back `parsed: null`.
bench.compact_line() is what is printed now: checked here on the last committed full record."""
import glob
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_compact_line_is_small_and_complete():
    b = _bench()
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_b32.json")))
    assert recs
    full = json.load(open(recs[-1]))
    if "roofline_unet_layers" not in full:        # a compact record (round 5 on): the detail file sits beside it
        full = json.load(open(recs[-1].replace(".json", "_detail.json")))
    dom = b.dominant_kernel_share()
    assert dom is not None and "same_library" in dom          # the committed profile is tied to the library it was measured on
    assert b.in_step_fraction(dict(dom, same_library=False), {"layers": full["roofline_unet_layers"]["layers"]}, 32) is None
    dom["same_library"] = True                                  # (this test is about the line, whatever library is built here)
    ins = b.in_step_fraction(dom, {"layers": full["roofline_unet_layers"]["layers"]}, 32)
    assert ins is not None and 0.05 < ins["frac"] < 1.0
    full["roofline"].update(in_step=ins, in_step_frac=ins["frac"], dominant_in_profile=dom)
    line = json.dumps(b.compact_line(full))
    assert len(line) < b.COMPACT_LIMIT and "\n" not in line
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert "workload" in back["config"] and "model" not in back["config"]
    r = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "in_step_frac", "step_frac"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    for k in ("roofline_wt_fwd", "roofline_wt_bwd"):
        assert back[k]["bound"] == "hbm" and back[k]["frac_of_copy"] > back[k]["frac"]
