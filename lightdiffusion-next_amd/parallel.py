"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no multi-device code at all.  The denoising path shards naturally: every latent
in a batch is denoised independently (GroupNorm / LayerNorm / attention are per-sample, calc_cond_batch only
concatenates, cond.py:150-288), so rank r takes a contiguous slice of the batch, keeps fully replicated weights,
and the only collective is ONE all-gather of the final latents per generation — no per-step traffic.
Noise keeps the reference's semantics: the whole batch is drawn once from the CPU generator
(ksampler_util.prepare_noise, ksampler_util.py:274-311) and then sliced, so results do not depend on the world size.
"""
import torch


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of `total` items for `rank`; the remainder goes to the first ranks."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_noise(latent_shape, seed: int, rank: int, world: int, dtype=torch.float32):
    """prepare_noise for the WHOLE batch (same RNG stream as the reference), then this rank's slice."""
    generator = torch.manual_seed(seed)
    noise = torch.randn(tuple(latent_shape), dtype=dtype, generator=generator, device="cpu")
    lo, hi = shard_bounds(latent_shape[0], rank, world)
    return noise[lo:hi]


def gather_latents(x_local: torch.Tensor, total: int, dist=None):
    """All-gather the per-rank final latents into the full batch [total, ...] on every rank (one collective)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return x_local
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (total + world - 1) // world                       # pad to equal chunks for all_gather
    pad = torch.zeros((per,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    pad[: x_local.shape[0]] = x_local
    chunks = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(chunks, pad)
    out = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        out.append(chunks[r][: hi - lo])
    return torch.cat(out, dim=0)


# ------------------------------------------------------------------------------------------------------------
# Direct RCCL path (round 4): the one collective of a generation without torch.distributed in the data path.
# ctypes on librccl.so (RCCL exports the NCCL API): ncclGetUniqueId on rank 0, the 128-byte id handed to the other ranks through a FILE
# (atomic rename; the launcher exports LDX_RCCL_ID_FILE) or any callable the caller supplies, ncclCommInitRank, then ncclAllGather on the
# caller's HIP stream straight from / into device pointers.  Same padding / slicing as gather_latents above, so both paths return equal tensors.
import ctypes as _C
import os as _os
import time as _time

NCCL_UNIQUE_ID_BYTES = 128
_NCCL_FLOAT32 = 7          # ncclDataType_t: ncclFloat32 (nccl.h)


class _NcclUniqueId(_C.Structure):
    # c_ubyte, not c_char: a c_char array field reads back NUL-terminated, and real ids hold zero bytes (magic, sockaddr, padding)
    _fields_ = [("internal", _C.c_ubyte * NCCL_UNIQUE_ID_BYTES)]


def unique_id_to_bytes(uid: "_NcclUniqueId") -> bytes:
    """All 128 bytes of the struct (bytes(struct) copies the whole buffer, zeros included)."""
    raw = bytes(uid)
    assert len(raw) == NCCL_UNIQUE_ID_BYTES, len(raw)
    return raw


def unique_id_from_bytes(raw: bytes) -> "_NcclUniqueId":
    if len(raw) != NCCL_UNIQUE_ID_BYTES:
        raise ValueError(f"RCCL unique id must be {NCCL_UNIQUE_ID_BYTES} bytes, got {len(raw)}")
    return _NcclUniqueId.from_buffer_copy(raw)


def load_rccl(path: str = None):
    """librccl.so with the four entry points' signatures set.  Raises OSError if the library is missing: there is no fallback."""
    lib = _C.CDLL(path or _os.environ.get("LDX_RCCL_LIB", "librccl.so"))
    lib.ncclGetUniqueId.argtypes = [_C.POINTER(_NcclUniqueId)]
    lib.ncclCommInitRank.argtypes = [_C.POINTER(_C.c_void_p), _C.c_int, _NcclUniqueId, _C.c_int]
    lib.ncclAllGather.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_size_t, _C.c_int, _C.c_void_p, _C.c_void_p]
    lib.ncclCommDestroy.argtypes = [_C.c_void_p]
    lib.ncclGetErrorString.argtypes = [_C.c_int]
    lib.ncclGetErrorString.restype = _C.c_char_p
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllGather, lib.ncclCommDestroy):
        f.restype = _C.c_int
    return lib


RUN_NONCE_BYTES = 16


def _run_nonce(nonce=None) -> bytes:
    """16 bytes that name THIS launch (the launcher exports LDX_RCCL_NONCE to every rank): a file left behind by an earlier or crashed run at the
    same path carries another nonce and is ignored by the readers instead of being taken for rank 0's id."""
    import hashlib
    text = nonce if nonce is not None else _os.environ.get("LDX_RCCL_NONCE", "")
    if not text:
        # plain torchrun (no bench.py self-launch): derive the name from what torchrun gives every rank of ONE launch — the rendezvous endpoint and run id —
        # so that a stale id file of a crashed run (other port / run id) can never be taken for this run's
        text = "|".join(_os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"))
    return hashlib.sha256(text.encode()).digest()[:RUN_NONCE_BYTES]


def exchange_unique_id_file(path: str, rank: int, make_id, timeout_s: float = 120.0, nonce: str = None) -> bytes:
    """Rank 0 calls make_id() -> 128 bytes and publishes nonce + id at `path` (stale file removed first; written to a temporary name, then renamed:
    readers never see a partial file); every other rank waits for a file that carries ITS run nonce.  Returns the id on every rank."""
    tag = _run_nonce(nonce)
    if rank == 0:
        uid = bytes(make_id())
        if len(uid) != NCCL_UNIQUE_ID_BYTES:
            raise ValueError(f"RCCL unique id must be {NCCL_UNIQUE_ID_BYTES} bytes, got {len(uid)}")
        try:
            _os.unlink(path)
        except FileNotFoundError:
            pass
        tmp = f"{path}.tmp.{_os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(tag + uid)
            f.flush()
            _os.fsync(f.fileno())
        _os.replace(tmp, path)
        return uid
    t0 = _time.monotonic()
    while True:
        try:
            with open(path, "rb") as f:
                blob = f.read()
            if len(blob) == RUN_NONCE_BYTES + NCCL_UNIQUE_ID_BYTES and blob[:RUN_NONCE_BYTES] == tag:
                return blob[RUN_NONCE_BYTES:]
        except FileNotFoundError:
            pass
        if _time.monotonic() - t0 > timeout_s:
            raise TimeoutError(f"RCCL unique id did not appear at {path} within {timeout_s} s")
        _time.sleep(0.01)


class RcclComm:
    """One RCCL communicator over the ranks of a node.  id_exchange(rank, make_id) -> bytes hands rank 0's unique id to everybody
    (default: the file named by LDX_RCCL_ID_FILE)."""

    def __init__(self, rank: int, world: int, id_exchange=None, lib=None):
        self.rank, self.world = rank, world
        self.lib = lib or load_rccl()

        def make_id():
            uid = _NcclUniqueId()
            self._check(self.lib.ncclGetUniqueId(_C.byref(uid)), "ncclGetUniqueId")
            return unique_id_to_bytes(uid)

        if id_exchange is None:
            path = _os.environ.get("LDX_RCCL_ID_FILE")
            if not path:
                raise RuntimeError("RcclComm: set LDX_RCCL_ID_FILE (a path every rank of the node can read) or pass id_exchange")
            id_exchange = lambda r, mk: exchange_unique_id_file(path, r, mk)
            id_path = path
        else:
            id_path = None
        self._id_path = id_path
        raw = id_exchange(rank, make_id)
        uid = unique_id_from_bytes(bytes(raw))
        self.comm = _C.c_void_p()
        self._check(self.lib.ncclCommInitRank(_C.byref(self.comm), world, uid, rank), "ncclCommInitRank")
        if rank == 0 and self._id_path:          # every rank has joined once rank 0's init returns: the id file has done its job
            try:
                _os.unlink(self._id_path)
            except OSError:
                pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.ncclGetErrorString(rc).decode()}")

    def all_gather_latents(self, x_local: torch.Tensor, total: int, stream_ptr: int = None) -> torch.Tensor:
        """gather_latents() through ncclAllGather: fp32 device tensors, equal padded chunks, then the true per-rank slices."""
        assert x_local.is_cuda and x_local.dtype == torch.float32
        per = (total + self.world - 1) // self.world
        pad = torch.zeros((per,) + tuple(x_local.shape[1:]), dtype=torch.float32, device=x_local.device)
        pad[: x_local.shape[0]] = x_local
        recv = torch.empty((self.world * per,) + tuple(x_local.shape[1:]), dtype=torch.float32, device=x_local.device)
        st = stream_ptr if stream_ptr is not None else torch.cuda.current_stream().cuda_stream
        self._check(self.lib.ncclAllGather(_C.c_void_p(pad.data_ptr()), _C.c_void_p(recv.data_ptr()), pad.numel(), _NCCL_FLOAT32, self.comm, _C.c_void_p(st)),
                    "ncclAllGather")
        return unpad_gathered(recv, total, self.world)

    def close(self):
        if getattr(self, "comm", None):
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


def unpad_gathered(recv: torch.Tensor, total: int, world: int) -> torch.Tensor:
    """[world * per, ...] of equal padded chunks -> the true [total, ...] batch (rank r contributed shard_bounds(total, r, world))."""
    per = recv.shape[0] // world
    out = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        out.append(recv[r * per: r * per + (hi - lo)])
    return torch.cat(out, dim=0)


# ---- rank -> GPU -> NUMA-node CPU affinity (round 4).  Eight ranks that each synthesise / convert 860 M parameters on "all" host threads serialise on
# one socket; each rank gets the cores of its GPU's NUMA node (sysfs), split evenly among the ranks that share the node.

def parse_cpulist(text: str):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def plan_rank_cpus(allowed, node_of_rank, node_cpus, local_rank: int):
    """CPUs for `local_rank`.  allowed: CPUs this process may use; node_of_rank: NUMA node per local rank (-1 / None = unknown);
    node_cpus: {node: [cpus]}.  Ranks with a known node share that node's allowed CPUs evenly (in rank order); ranks without one share
    `allowed` evenly over all local ranks.  Never returns an empty list."""
    allowed = sorted(set(allowed))
    world = len(node_of_rank)
    node = node_of_rank[local_rank]
    pool = sorted(set(node_cpus.get(node, [])) & set(allowed)) if node is not None and node >= 0 else []
    if pool:
        peers = [r for r in range(world) if node_of_rank[r] == node]
    else:
        pool, peers = allowed, list(range(world))
    i, n = peers.index(local_rank), len(peers)
    lo, hi = i * len(pool) // n, (i + 1) * len(pool) // n
    return pool[lo:hi] if hi > lo else [pool[i % len(pool)]]


def masked_rank_cpus(allowed, my_node: int, node_cpus, local_rank: int, local_world: int):
    """CPUs for a rank that only sees its OWN GPU (per-rank HIP_VISIBLE_DEVICES): the peers' nodes are unknown, so nothing is guessed about them.  The rank's
    node (my_node, -1 = unknown) is cut into per_node = ceil(local_world / nodes) equal slices and the rank takes slice local_rank % per_node: with the usual
    even, rank-ordered spread (ranks [k * per_node, (k + 1) * per_node) on node k) the slices of one node's ranks are disjoint by construction — every rank computes
    the same cut from the same inputs, whatever it believes about the others.  Unknown node: `allowed` cut into local_world slices.  Never empty."""
    allowed = sorted(set(allowed))
    pool = sorted(set(node_cpus.get(my_node, [])) & set(allowed)) if my_node is not None and my_node >= 0 else []
    if pool:
        n = -(-local_world // max(1, len(node_cpus)))
        i = local_rank % n
    else:
        pool, n, i = allowed, local_world, local_rank
    lo, hi = i * len(pool) // n, (i + 1) * len(pool) // n
    return pool[lo:hi] if hi > lo else [pool[i % len(pool)]]


def gpu_numa_node(device_index: int) -> int:
    """NUMA node of a visible GPU from sysfs (PCI address from the device properties); -1 when the platform does not say or the index is not
    visible to THIS process (per-rank HIP_VISIBLE_DEVICES: every rank only sees its own GPU as device 0)."""
    try:
        if device_index < 0 or device_index >= torch.cuda.device_count():
            return -1
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            return int(f.read().strip())
    except Exception:
        return -1


def bind_rank_to_numa(local_rank: int, local_world: int):
    """Restrict this process to its share of the cores next to its GPU; returns the CPU list (also the torch thread count to use).
    local_world = ranks on THIS node (LOCAL_WORLD_SIZE).  When the ranks were given one visible GPU each (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES),
    the peers' GPUs cannot be queried from here: this rank's own node is looked up as device 0 and the node's cores are split among all local ranks
    that could share it (an even split of the node: never worse than the un-bound default, never an empty set)."""
    import glob
    import os
    allowed = sorted(os.sched_getaffinity(0))
    nodes = {}
    for d in glob.glob("/sys/devices/system/node/node[0-9]*"):
        try:
            with open(os.path.join(d, "cpulist")) as f:
                nodes[int(os.path.basename(d)[4:])] = parse_cpulist(f.read())
        except Exception:
            pass
    masked = any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")) and torch.cuda.device_count() < local_world
    if masked:
        cpus = masked_rank_cpus(allowed, gpu_numa_node(0), nodes, local_rank, local_world)
    else:
        cpus = plan_rank_cpus(allowed, [gpu_numa_node(r) for r in range(local_world)], nodes, local_rank)
    try:
        os.sched_setaffinity(0, cpus)
    except Exception:
        pass
    return cpus


def shared_state_dict(dist, spec, rank: int, world: int, local_rank: int, local_world: int, seed: int = 1234, dirs=("/dev/shm", "/tmp"), tag: str = "0"):
    """ONE synthesis of the synthetic state dict per node (bench.py, N > 1): local rank 0 writes it under the first of `dirs` with enough free space
    (weights.publish_state_dict), the other ranks of the node map it copy-on-write, local rank 0 unlinks it.  A tmpfs that is too small would kill the writer
    with SIGBUS (not an exception) and leave the others waiting, so the free space is checked first, and EVERY rank learns the outcome (one
    all_gather_object) before it touches the file; if anything fails, every rank of that node synthesises privately.  Returns (state dict, description)."""
    import os
    from . import weights
    need = int(weights.state_dict_layout(spec)[1] * 1.02) + (1 << 20)
    path, why, sd = None, "no directory with enough free space", None
    if local_rank == 0:
        for d in dirs:
            try:
                st = os.statvfs(d)
                if os.path.isdir(d) and os.access(d, os.W_OK) and st.f_bavail * st.f_frsize >= need:
                    path = os.path.join(d, f"ldx_sd_{tag}_{os.getppid()}_{rank}.bin")
                    break
            except OSError:
                continue
        if path:
            try:
                sd = weights.publish_state_dict(spec, path, seed=seed)
            except Exception as e:                               # noqa: BLE001 - any failure means "fall back"
                why, path, sd = f"{type(e).__name__}: {e}", None, None
    got = [None] * world
    dist.all_gather_object(got, (rank, local_rank, path, why))
    node = rank // max(local_world, 1)
    mine = [g for g in got if g[1] == 0 and g[0] // max(local_world, 1) == node]      # this node's local rank 0
    path, why = (mine[0][2], mine[0][3]) if mine else (None, "no local rank 0 on this node")
    failed = False
    if path and local_rank != 0:
        try:
            sd = weights.attach_state_dict(spec, path)
        except Exception as e:                                   # noqa: BLE001
            sd, why, failed = None, f"attach failed: {e}", True
    dist.barrier()                                               # everybody has mapped (or given up): the name can go, the mappings stay valid
    if path and local_rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass
    if sd is None:
        return weights.synth_state_dict(spec, seed=seed), f"private synthesis (shared state dict unavailable: {why})"
    return sd, f"shared: local rank 0 synthesised, the others mapped {os.path.dirname(path)}"
