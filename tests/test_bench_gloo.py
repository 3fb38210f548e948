"""bench.py's own N > 1 control flow on CPU: `python bench.py --gpus 2 --stub-engine` must self-launch two gloo ranks
(torch.distributed.run), shard ONE prepare_noise draw, time the regions, all-gather once and print ONE JSON line with
n_gpus = 2 — and the gathered latents must equal the single-process run of the same global batch bit for bit.
The stub engine is an analytic per-sample stand-in for the UNet (bench.StubEngine); no HIP code runs here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, check=True):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--stub-engine", "--latent", "8", "--steps", "3", "--warmup", "1", "--repeats", "2"] + extra
    p = None
    for attempt in range(3):          # the self-launch picks a free rendezvous port and releases it before torchrun binds it: a lost race is retried, not reported
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            if attempt == 2:
                raise
            continue
        if p.returncode == 0 or not check or not any(t in p.stderr for t in ("Address already in use", "EADDRINUSE", "address already in use")):
            break
    if check:
        assert p.returncode == 0, p.stderr[-2000:]
    return p


def _line(p):
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_config3_matches_single_process():
    one = _line(_run(["--gpus", "1", "--config", "3", "--global-batch", "6"]))
    two = _line(_run(["--gpus", "2", "--config", "3", "--global-batch", "6"]))
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["timing"]["rccl_ranks"] == 2 and two["timing"]["backend"] == "gloo"
    assert two["scaling"] == "strong" and two["config"]["images_per_gpu"] == 3 and two["config"]["global_batch"] == 6
    assert len(two["timing"]["per_rank_ms_per_step"]) == 2 and len(two["timing"]["region_ms_per_step"]) == 2
    assert one["timing"]["latents_sha256_16"] == two["timing"]["latents_sha256_16"]
    assert two["metric"].startswith("STUB") and two["vs_baseline"] is None


def test_weak_scaling_mode_several_images_per_rank():
    # the round-1 crash: --batch > 1 with N > 1 (gather_latents was sized with the world size, not the item count)
    two = _line(_run(["--gpus", "2", "--batch", "2"]))
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["global_batch"] == 4
    one = _line(_run(["--gpus", "1", "--batch", "4"]))
    assert one["timing"]["latents_sha256_16"] == two["timing"]["latents_sha256_16"]
    assert two["value"] > 0 and abs(two["value"] - 2 * 3 / (two["ms_per_step"] * 3e-3)) / two["value"] < 1e-2


def test_refuses_world_size_mismatch():
    p = _run(["--gpus", "2"], env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"}, check=False)
    assert p.returncode != 0 and "refusing to mis-report" in p.stderr


def test_eight_ranks_config3_even_and_uneven_global_batch():
    """The north_star's 8-GPU leg as far as a GPU-less box allows: bench.py's own 8-rank launch (gloo), SURVEY config 3 with the global batch
    drawn once and sharded 8 ways — evenly (64 -> 8 per rank) and unevenly (10 -> 2,2,1,1,1,1,1,1) — must reproduce the single-process latents."""
    for gb in (64, 10):
        one = _line(_run(["--gpus", "1", "--config", "3", "--global-batch", str(gb)]))
        eight = _line(_run(["--gpus", "8", "--config", "3", "--global-batch", str(gb)]))
        assert eight["n_gpus"] == 8 and eight["timing"]["rccl_ranks"] == 8 and len(eight["timing"]["per_rank_ms_per_step"]) == 8
        assert eight["config"]["global_batch"] == gb and eight["config"]["images_per_gpu"] == -(-gb // 8) and eight["scaling"] == "strong"
        assert eight["timing"]["allgather_bytes"] == gb * 4 * 8 * 8 * 4
        assert one["timing"]["latents_sha256_16"] == eight["timing"]["latents_sha256_16"], gb


def test_preflight_two_ranks_prints_topology_and_exits():
    """`bench.py --gpus N --preflight` (round 5): process group + one 256 KiB all-gather + per-rank device info, no weight work; and the normal
    N > 1 run carries the same object in `timing.preflight` (it runs before the engine is built)."""
    p = _run(["--gpus", "2", "--preflight"])
    pf = _line(p)
    assert pf["n_gpus"] == 2 and pf["preflight"]["world"] == 2 and pf["preflight"]["backend"] == "gloo"
    assert pf["preflight"]["allgather_256KiB_ok"] is True
    assert [r["rank"] for r in pf["preflight"]["ranks"]] == [0, 1] and pf["preflight"]["ranks"][0]["pid"] != pf["preflight"]["ranks"][1]["pid"]
    two = _line(_run(["--gpus", "2", "--batch", "1"]))
    assert two["timing"]["preflight"]["world"] == 2 and len(two["timing"]["engine_build"]) == 2
