"""CPU: `python bench.py --gpus N` (N > 1) without a torchrun environment starts its own N rank processes (bench.launch_ranks)
and relays rank 0's result line and the ranks' exit status.  The real rank program needs a GPU, so the launcher is run
here with a stub child that does what a rank does around the measurement: rendezvous from the torchrun environment on
127.0.0.1 (gloo), a collective, one JSON line from rank 0."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent('''
    import json, os, sys
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ.get("EMB_BENCH_LAUNCHED") == str(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1.0]); dist.all_reduce(t)
    if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        sys.exit(7)
    print(f"noise from rank {rank}", flush=True)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": float(t.item()), "n_gpus": world, "argv": sys.argv[1:],
                          "ranks_seen": dist.get_world_size()}), flush=True)
    dist.barrier(); dist.destroy_process_group()
''')


def _run(tmp_path, extra):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(STUB)
    drv = tmp_path / "drive.py"
    drv.write_text(f"import sys; sys.path.insert(0, {ROOT!r}); import bench\n"
                   f"sys.exit(bench.launch_ranks(2, ['--gpus', '2'] + {extra!r}, script={str(stub)!r}, timeout=240))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, env=env, timeout=300)


def test_self_launch_relays_rank0_line_and_exit_code(tmp_path):
    r = _run(tmp_path, ["--steps", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                              # ONE JSON line on stdout; everything else went to stderr
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["value"] == 3.0
    assert out["argv"] == ["--gpus", "2", "--steps", "3"]        # the ranks get the parent's arguments unchanged
    assert "noise from rank" in r.stderr


def test_self_launch_reports_a_failed_rank(tmp_path):
    r = _run(tmp_path, ["--fail-rank", "1"])
    assert r.returncode != 0                                      # a failed rank fails the run; nothing is retried


def test_launcher_decision_is_taken_before_torch_is_imported():
    """`bench.py --gpus 2` as the driver starts N = 1: the parent must hand over to child ranks without importing torch
    (nothing in the parent may initialise a device); with a torchrun environment, or --gpus 1, it must not launch."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("_launch_if_needed(sys.argv[1:])") < src.index("\nimport torch")
    probe = ("import sys, os; sys.path.insert(0, %r); sys.argv = ['bench.py'] + %%r\n"
             "import bench\n"
             "bench.launch_ranks = lambda n, argv, **k: (print('LAUNCH', n, 'torch' in sys.modules), 0)[1]\n"
             "import importlib; sys.modules.pop('torch', None)\n"
             "try:\n    bench._launch_if_needed(sys.argv[1:])\n    print('NO LAUNCH')\nexcept SystemExit as e:\n    print('EXIT', e.code)\n") % ROOT
    def run(argv, env_extra=None):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        env.update(env_extra or {})
        return subprocess.run([sys.executable, "-c", probe % (argv,)], capture_output=True, text=True, env=env, timeout=120).stdout
    assert "LAUNCH 2" in run(["--gpus", "2"]) and "EXIT 0" in run(["--gpus", "2"])
    assert "LAUNCH 4" in run(["--gpus=4", "--steps", "5"])
    assert "NO LAUNCH" in run(["--gpus", "1"])
    assert "NO LAUNCH" in run(["--gpus", "2"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert "NO LAUNCH" in run(["--gpus", "2", "--force-collectives"])
