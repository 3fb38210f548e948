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
