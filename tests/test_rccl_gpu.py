"""The multi-GPU data path's two collectives meet the REAL RCCL on this box's one GPU (world size 1): no scaling statement, but the first hardware run of the
N > 1 path is then no longer the first run of the ctypes signatures, the 128-byte unique id, the stream hand-over and the nccl backend's bring-up.
(The 8-rank semantics are covered on gloo / a fake library: tests/test_parallel_gloo.py, test_bench_gloo.py, test_rccl_ctypes_cpu.py.)  Subprocesses with a
timeout: a communicator that hangs must fail this test, not the suite."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, timeout=240):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code) % ROOT], env=env, capture_output=True, text=True, timeout=timeout)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_direct_rccl_path_on_one_rank():
    out = _run('''
        import sys, torch
        sys.path.insert(0, %r)
        import ldx_amd as ldx
        par = ldx.parallel
        comm = par.RcclComm(0, 1, id_exchange=lambda r, make_id: make_id())        # librccl.so: ncclGetUniqueId -> ncclCommInitRank
        g = torch.Generator().manual_seed(1)
        for shape in ((1, 4, 128, 128), (3, 4, 64, 64)):
            x = torch.randn(shape, generator=g).cuda()
            got = comm.all_gather_latents(x, shape[0])                               # ncclAllGather on torch's current stream
            torch.cuda.synchronize()
            assert got.shape == x.shape and torch.equal(got, x), shape
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            y = torch.randn(2, 4, 32, 32, generator=g).cuda()
            got = comm.all_gather_latents(y, 2, stream_ptr=s.cuda_stream)
        s.synchronize()
        assert torch.equal(got, y)
        comm.close()
        print("RCCL_DIRECT_OK")
    ''')
    assert "RCCL_DIRECT_OK" in out


def test_torch_distributed_nccl_backend_on_one_rank():
    out = _run('''
        import os, socket, sys, torch
        sys.path.insert(0, %r)
        import torch.distributed as dist
        import ldx_amd as ldx
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        x = torch.randn(2, 4, 64, 64, device="cuda")
        chunks = [torch.empty_like(x)]
        dist.all_gather(chunks, x); torch.cuda.synchronize()
        assert torch.equal(chunks[0], x)
        assert torch.equal(ldx.parallel.gather_latents(x, 2, dist), x)
        noise = ldx.parallel.shard_noise((2, 4, 64, 64), 42, 0, 1)
        assert noise.shape[0] == 2
        dist.barrier(); dist.destroy_process_group()
        print("NCCL_PG_OK", dist.is_nccl_available())
    ''')
    assert "NCCL_PG_OK" in out


def test_bench_preflight_on_one_gpu():
    """bench.py --gpus 1 --preflight: the topology line (device, NUMA node) without any weight work."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--preflight"], capture_output=True, text=True, timeout=300)
    print(r.stdout[-800:], r.stderr[-800:])
    assert r.returncode == 0 and '"preflight"' in r.stdout and '"n_gpus": 1' in r.stdout
