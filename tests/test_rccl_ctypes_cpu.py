"""Direct RCCL path (lightdiffusion-next_amd/parallel.py, round 4) as far as a GPU-less box can take it: librccl.so loads through ctypes and exports
the four entry points; rank 0's unique id reaches two other PROCESSES through the file hand-off (a fake id maker stands in for ncclGetUniqueId where
that needs a device); the gathered-chunk bookkeeping equals gather_latents' on even and uneven batches.  The collective itself needs >= 2 GPUs and is
exercised by `bench.py --gpus N --rccl-direct` on the driver's multi-GPU node."""
import multiprocessing as mp
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_librccl_exports_the_entry_points(ldx):
    try:
        lib = ldx.parallel.load_rccl()
    except OSError as e:
        pytest.skip(f"librccl.so not loadable here: {e}")
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllGather", "ncclCommDestroy", "ncclGetErrorString"):
        assert hasattr(lib, name)


def _reader(rank, path, q):
    sys.path.insert(0, ROOT)
    import ldx_amd as ldx
    q.put((rank, ldx.parallel.exchange_unique_id_file(path, rank, make_id=None, timeout_s=30.0)))


def test_unique_id_file_handoff_across_processes(ldx, tmp_path):
    path = str(tmp_path / "rccl_id")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reader, args=(r, path, q)) for r in (1, 2)]
    for p in procs:
        p.start()
    uid = bytes(range(128))
    got0 = ldx.parallel.exchange_unique_id_file(path, 0, make_id=lambda: uid)
    res = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(30)
    assert got0 == uid and res == {1: uid, 2: uid}
    with pytest.raises(TimeoutError):
        ldx.parallel.exchange_unique_id_file(str(tmp_path / "never"), 1, make_id=None, timeout_s=0.2)


@pytest.mark.parametrize("total,world", [(8, 2), (10, 8), (5, 2), (64, 8)])
def test_unpad_matches_shard_bounds(ldx, total, world):
    per = (total + world - 1) // world
    full = torch.arange(total * 3, dtype=torch.float32).reshape(total, 3)
    chunks = []
    for r in range(world):
        lo, hi = ldx.parallel.shard_bounds(total, r, world)
        pad = torch.full((per, 3), -1.0)
        pad[: hi - lo] = full[lo:hi]
        chunks.append(pad)
    assert torch.equal(ldx.parallel.unpad_gathered(torch.cat(chunks), total, world), full)
