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


def test_unique_id_with_nul_bytes_round_trips_through_the_struct(ldx):
    """Real ncclUniqueIds hold zero bytes (magic, sockaddr, padding): the struct <-> bytes conversion must keep all 128 (a c_char array
    field reads back NUL-terminated and cut the id at its first zero)."""
    par = ldx.parallel
    raw = bytes([7, 0, 0, 9] + [0] * 60 + list(range(64)))
    assert len(raw) == 128
    uid = par.unique_id_from_bytes(raw)
    assert par.unique_id_to_bytes(uid) == raw
    with pytest.raises(ValueError):
        par.unique_id_from_bytes(raw[:5])

    class FakeLib:                       # ncclGetUniqueId writes an id with NULs; ncclCommInitRank must receive exactly those bytes
        seen = None
        def ncclGetUniqueId(self, p):
            import ctypes
            ctypes.memmove(p, raw, 128); return 0
        def ncclCommInitRank(self, comm, world, uid, rank):
            FakeLib.seen = bytes(uid); return 0
        def ncclCommDestroy(self, c): return 0
        def ncclGetErrorString(self, rc): return b"fake"
    comm = par.RcclComm(0, 1, id_exchange=lambda r, mk: mk(), lib=FakeLib())
    assert FakeLib.seen == raw
    comm.close()


def test_stale_id_file_of_another_run_is_ignored(ldx, tmp_path):
    par = ldx.parallel
    path = str(tmp_path / "id")
    old = bytes(range(128))
    assert par.exchange_unique_id_file(path, 0, make_id=lambda: old, nonce="run-A") == old
    with pytest.raises(TimeoutError):                       # a reader of run B must not take run A's file
        par.exchange_unique_id_file(path, 1, make_id=None, timeout_s=0.3, nonce="run-B")
    new = bytes(reversed(range(128)))
    assert par.exchange_unique_id_file(path, 0, make_id=lambda: new, nonce="run-B") == new
    assert par.exchange_unique_id_file(path, 1, make_id=None, timeout_s=5, nonce="run-B") == new


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
