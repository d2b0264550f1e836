"""bench.py's own launcher (CPU tier): `python bench.py --gpus N` with no launcher environment must start N ranks as a CHILD
(python -m torch.distributed.run ... bench.py <same args>) before anything in the parent touches HIP, and relay its line."""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_argv_is_the_drivers_launch_line():
    import bench
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.spawn_argv(argv, 8, 29611)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == argv          # the child sees exactly the parent's flags


def test_presets_name_the_baseline_configs():
    import bench
    a = bench.parse_args(["--preset", "config5", "--gpus", "8"])
    assert a.samples == 1 << 33 and a.gpus == 8 and a.channels == 1      # 16 GiB of i8 IQ per GPU
    a = bench.parse_args(["--preset", "config4"])
    assert a.channels == 64 and a.samples == 1 << 29
    a = bench.parse_args([])
    assert a.samples == 1 << 29 and a.gpus == 1 and a.channels == 1      # BASELINE configs[1]: the metric's workload
    a = bench.parse_args(["--preset", "config3", "--samples", "4096"])   # an explicit flag wins
    assert a.samples == 4096


def test_nothing_touches_the_gpu_before_the_spawn():
    """The self-launch block must come before `import torch` / `import air_rs_amd` in bench.py: a parent that has initialised
    HIP and then starts ranks is what the GPU boxes forbid."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    first_heavy = min(n.lineno for n in ast.walk(tree) if isinstance(n, (ast.Import, ast.ImportFrom)) and
                      any((a.name if isinstance(n, ast.Import) else (n.module or "")).split(".")[0] in ("torch", "air_rs_amd", "numpy")
                          for a in n.names) and n.col_offset == 0)
    launch = next(n.lineno for n in tree.body if isinstance(n, ast.If) and "self_launch" in ast.unparse(n))
    assert launch < first_heavy, (launch, first_heavy)
    assert "subprocess.run" in src and "os.exec" not in src            # a child, never an exec
