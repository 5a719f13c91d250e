"""CPU: `bench.py --gpus N` never reports a world other than the one asked for (round-4 review: started WITHOUT a launcher it
measured one GPU and printed "n_gpus": 1).  Without WORLD_SIZE it starts its N ranks itself through torch.distributed.run
and relays rank 0's line; a WORLD_SIZE that disagrees with --gpus, or fewer visible devices than ranks, exits non-zero.
The rendezvous-only mode (--launch-check, gloo) exercises the self-launcher end to end on a box without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def _json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{"):
            out.append(json.loads(line))
    return out


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = _run(["--gpus", "2", "--launch-check", "--backend", "gloo", "--one-device"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0] == {"launch_check": True, "n_gpus": 2, "backend": "gloo"}, r.stdout


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = _run(["--gpus", "2", "--launch-check", "--backend", "gloo"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and not _json_lines(r.stdout)
    assert "WORLD_SIZE=1" in r.stderr
    r = _run(["--gpus", "1", "--launch-check", "--backend", "gloo"], env_extra={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and not _json_lines(r.stdout)


def test_more_ranks_than_devices_is_refused_before_anything_is_measured():
    import torch
    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--cpu-events", "0"])
    assert r.returncode != 0 and not _json_lines(r.stdout), r.stdout
    assert "device(s) visible" in r.stderr


def test_single_rank_launch_check_prints_the_world_it_ran():
    r = _run(["--gpus", "1", "--launch-check", "--backend", "gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout) == [{"launch_check": True, "n_gpus": 1, "backend": "gloo"}]
