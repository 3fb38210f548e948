"""The N > 1 path on CPU: two gloo processes shard a batch, run a per-sample 'denoiser' (the oracle's sampler
loop over a cheap analytic model — per-sample independence is what matters), all-gather once, and must reproduce
the single-process result bit for bit, for even and uneven batch sizes."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_workers(target, world, args_of_rank, n_results, seconds=150, attempts=3):
    """Spawn `world` worker processes on a fresh rendezvous port and collect `n_results` results.  A blocking SimpleQueue.get() waited forever when a worker died
    before it could report (a rendezvous port taken between _free_port() and the store's bind: seen once per ~10 runs of the CPU suite in the build container), so
    the wait is bounded and a silent attempt is repeated on another port; only `attempts` silent attempts in a row fail the test."""
    import queue
    ctx = mp.get_context("spawn")
    codes = None
    for _ in range(attempts):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args_of_rank(r, port)) + (q,)) for r in range(world)]
        for p in procs:
            p.start()
        out = []
        try:
            for _i in range(n_results):
                out.append(q.get(timeout=seconds))
        except queue.Empty:
            out = None
        for p in procs:
            p.join(timeout=30 if out is None else 120)
            if p.is_alive():
                p.terminate()
                p.join(timeout=30)
        codes = [p.exitcode for p in procs]
        if out is not None:
            assert all(c == 0 for c in codes), codes
            return out
    pytest.fail(f"no result from the worker processes in {attempts} attempts of {seconds} s (last exit codes {codes})")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _toy_sample(noise):
    """Per-sample work with the same structure as the sampler loop (sigma schedule + Euler update)."""
    sys.path.insert(0, ROOT)
    from oracle import sd15_oracle as O
    sig = O.calculate_sigmas("karras", 6)
    x = noise * torch.sqrt(1.0 + sig[0] ** 2)
    for i in range(len(sig) - 1):
        den = torch.tanh(x) * (1.0 / (1.0 + sig[i]))          # stands in for the UNet: strictly per-sample
        x = x + ((x - den) / sig[i]) * (sig[i + 1] - sig[i])
    return x


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import ldx_amd as ldx
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    noise = ldx.parallel.shard_noise((total, 4, 8, 8), 42, rank, world)
    lo, hi = ldx.parallel.shard_bounds(total, rank, world)
    assert noise.shape[0] == hi - lo
    out = ldx.parallel.gather_latents(_toy_sample(noise), total, dist)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5])
def test_batch_shard_matches_single_process(ldx, total):
    world = 2
    got = _run_workers(_worker, world, lambda r, port: (total,), 1)[0]
    want = _toy_sample(ldx.parallel.shard_noise((total, 4, 8, 8), 42, 0, 1))
    assert got.shape == want.shape and torch.equal(got, want)


def test_shard_bounds_cover(ldx):
    for total in (1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [ldx.parallel.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def _sd_worker(rank, world, port, path, q):
    """bench.py's start-up protocol for the shared synthetic state dict: local rank 0 publishes, barrier, the others map, barrier, rank 0 unlinks."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import ldx_amd as ldx
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = ldx.UNetConfig.tiny(64, 128)
    spec = ldx.weights.unet_state_dict_spec(cfg)
    # the function bench.py calls (free-space check, outcome agreed by all ranks, unlink); `path` is the directory to publish in — or one that does
    # not exist, which must make BOTH ranks fall back to private synthesis instead of leaving rank 1 waiting for a file
    sd, how = ldx.parallel.shared_state_dict(dist, spec, rank, world, rank, world, seed=1234, dirs=(path,), tag=str(port))
    assert how.startswith("shared") == os.path.isdir(path), how
    assert not [f for f in (os.listdir(path) if os.path.isdir(path) else []) if f.startswith("ldx_sd_")], "the published file must be unlinked"
    want = ldx.weights.synth_state_dict(spec, seed=1234)
    ok = set(sd) == set(want) and all(sd[k].dtype == want[k].dtype and torch.equal(sd[k], want[k]) for k in want)
    if rank != 0:                       # a rank's writes must stay private (copy-on-write mapping)
        sd[spec[0][0]].zero_()
    ok = ok and (how.startswith("shared") or "unavailable" in how)
    dist.barrier()
    ok = ok and (rank != 0 or torch.equal(sd[spec[0][0]], want[spec[0][0]]))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("usable", [True, False])
def test_shared_state_dict_equals_private_synthesis(ldx, tmp_path, usable):
    """Round 5 (multi-GPU start-up): one synthesis per node through a mapped file == every rank's own synth_state_dict, bit for bit; and when no
    directory can take the file (usable = False) every rank falls back to its own synthesis — nobody waits for a file that will not come."""
    world = 2
    d = tmp_path / ("pub" if usable else "missing")
    if usable:
        d.mkdir()
    res = dict(_run_workers(_sd_worker, world, lambda r, port: (str(d),), world))
    assert res == {0: True, 1: True}
    path = str(tmp_path / "sd.bin")
    with pytest.raises(RuntimeError):                                   # a file of another model is refused, not mis-mapped
        spec = ldx.weights.unet_state_dict_spec(ldx.UNetConfig.tiny(64, 128))
        with open(path, "wb") as f:
            f.write(b"\0" * 128)
        ldx.weights.attach_state_dict(spec, path)
