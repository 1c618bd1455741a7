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


def test_bare_multi_gpu_invocation_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 8` started bare (as the driver starts N = 1) died on an assert.  With WORLD_SIZE unset it
    now runs the N ranks itself as a CHILD torch.distributed.run — before anything in the parent touches a GPU — forwards its flags,
    relays the child's stdout and exits with its status.  Checked up to the launch decision with a fake launcher (no GPU here)."""
    import subprocess
    import sys
    fake = tmp_path / "fake_launcher.py"
    fake.write_text("import json, sys\nprint(json.dumps({'argv': sys.argv[1:]}))\nsys.exit(7)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["WTPSE_BENCH_LAUNCHER"] = "%s %s" % (sys.executable, fake)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 7, (r.returncode, r.stderr[-500:])
    argv = json.loads(r.stdout.strip().splitlines()[-1])["argv"]
    assert argv[:3] == ["--nnodes=1", "--nproc-per-node", "2"]
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and int(argv[argv.index("--master-port") + 1]) > 0
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    # the real launcher is torch.distributed.run, one process per GPU
    b = _bench()
    cmd = b.self_launch_command(["--gpus", "4"], 4, port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
