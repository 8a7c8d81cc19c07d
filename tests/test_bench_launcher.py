"""bench.py --gpus N starts N ranks itself (fresh children before any GPU call).  Exercised on CPU with the gloo
backend and --dry (launcher, rendezvous, frame sharding, PaddedGather pipeline, barriers, max-over-ranks timing, the
single JSON line); the engine itself is not involved."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_gpus2_launches_two_ranks_weak():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2", "--warmup", "0", "--batch", "4", "--inner", "3"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dry"] is True
    assert d["config"]["frames_per_step_total"] == 4 * 3 * 2 and d["steps"] == 2


def test_frames_mode_is_config3_strong_split():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry", "--steps", "1", "--warmup", "0", "--frames", "7"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["shard_sizes"] == [3, 4] and d["config"]["frames_per_step_total"] == 7


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry"], {"RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and b"WORLD_SIZE" in r.stderr


def test_under_an_external_launcher_no_second_spawn():
    # the driver's form: python -m torch.distributed.run ... bench.py --gpus 2 (RANK set -> no spawning of its own)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--backend", "gloo", "--dry", "--steps", "1", "--warmup", "0", "--batch", "2", "--inner", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_a_rank_that_dies_early_ends_the_run_quickly():
    """Rank 1 exits at start-up while rank 0 waits in the rendezvous: the launcher notices, ends rank 0 and fails within
    seconds instead of sitting in init_process_group until torch's timeout."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry", "--steps", "1", "--warmup", "0", "--batch", "2", "--inner", "1",
              "--fail-rank", "1"], timeout=120)
    assert r.returncode == 1 and time.time() - t0 < 60, (r.returncode, time.time() - t0)
    assert b"rank 1 exited with code 7" in r.stderr and b"rank exit codes" in r.stderr


def test_launcher_deadline():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry", "--steps", "1", "--warmup", "0", "--frames", "64", "--min-region-s", "30"],
             {"BRISK_BENCH_DEADLINE_S": "4"}, timeout=120)
    assert r.returncode == 1 and b"deadline" in r.stderr


def test_frames_mode_repeats_the_step_until_the_region_is_long_enough():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2", "--warmup", "0", "--frames", "16", "--min-region-s", "0.5"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["config"]["timed_region_s"] >= 0.45 and d["config"]["steps_effective"] >= 2 and d["config"]["steps_effective"] % 2 == 0
