"""bench.py's `--gpus N` contract on the CPU side: N means N.  Under a launcher it must equal WORLD_SIZE, run plainly with
N > 1 the script starts its own ranks (torch.distributed.run on 127.0.0.1) before it touches a GPU, and a line with
n_gpus != N is never printed (round-3 verdict: `--gpus 8` run plainly measured one GPU and said n_gpus: 1, rc 0)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (import has no side effects: everything is under main())


def test_resolve_world_rules():
    assert bench.resolve_world(1, {}) == ("run", 0, 0, 1)
    assert bench.resolve_world(8, {}) == ("launch", 8)
    assert bench.resolve_world(4, {"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2"}) == ("run", 2, 2, 4)
    assert bench.resolve_world(1, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}) == ("run", 0, 0, 1)
    for gpus, env in ((8, {"WORLD_SIZE": "2"}), (1, {"WORLD_SIZE": "2"}), (0, {}), (2, {"WORLD_SIZE": "2", "RANK": "5"}),
                      (2, {"WORLD_SIZE": "two"})):
        assert bench.resolve_world(gpus, env)[0] == "error", (gpus, env)


def test_launch_command_is_the_drivers():
    cmd = bench.launch_command(["--gpus", "4", "--steps", "20"], 4, 29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "20"]


def test_mismatch_exits_2_without_a_line():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and p.stdout == "" and "WORLD_SIZE=4" in p.stderr
