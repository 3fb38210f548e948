"""Host-side mirror of the reference's scheduler loop (src/sample/*, src/cond/cond.py) driving libldx.

Same names, argument meaning and quirks as the reference so parity tests read like
reference code:
  ModelSamplingDiscrete       src/sample/sampling.py:221-356
  calculate_sigmas & friends  src/sample/ksampler_util.py:152-271, sampling_util.py:106-125
  prepare_noise               src/sample/ksampler_util.py:274-311   (CPU RNG, identical stream)
  calc_cond_batch             src/cond/cond.py:150-288               ([uncond; cond] batch, lcm-padded ctx)
  cfg / sampling_function     src/sample/CFG.py:6-161
  sample_euler                src/sample/samplers.py:166-327         (incl. multiscale)
  sample_dpmpp_2m_cfgpp       src/sample/samplers.py:754-962         (degenerates to 1st-order DPM++, A-2)
  KSampler.sample / sample1   src/sample/sampling.py:773-1045        (multiscale whitelist quirk, A-3)
Latents stay on the GPU in fp32; the per-step update runs in the HIP kernels behind ldx_sampler_step /
ldx_bilinear.  Only Python scalars (sigmas) live on the host.
"""
import math
import os

import numpy as np
import torch

from . import lib
from .engine import sd15_sigmas


# ------------------------------------------------------------------------------------------------------
class ModelSamplingDiscrete:
    """sampling.py:221-356 for the SD1.5 'linear' schedule + EPS (sampling.py:26-97)."""

    sigma_data = 1.0

    def __init__(self):
        self.sigmas, self.log_sigmas = sd15_sigmas()

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def timestep(self, sigma):
        log_sigma = sigma.log()
        dists = log_sigma - self.log_sigmas[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)

    def sigma(self, timestep):
        t = torch.clamp(timestep.float(), min=0, max=(len(self.sigmas) - 1))
        low_idx, high_idx, w = t.floor().long(), t.ceil().long(), t.frac()
        log_sigma = (1 - w) * self.log_sigmas[low_idx] + w * self.log_sigmas[high_idx]
        return log_sigma.exp()


def flux_time_shift(mu, sigma, t):
    """sampling.flux_time_shift (sampling.py:158-169)."""
    return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)


class ModelSamplingFlux:
    """sampling.ModelSamplingFlux + CONST (sampling.py:100-218): 10000-entry shifted sigma table, timestep(sigma) = sigma,
    x0 = sigma*noise + (1-sigma)*latent.  Like the reference's class it has NO sigma_min, so the "normal" and "karras"
    schedulers raise AttributeError for Flux exactly as they do there; "simple" and "beta" only read the table."""

    def __init__(self, shift=1.15, timesteps=10000):
        self.shift = shift
        self.sigmas = self.sigma(torch.arange(1, timesteps + 1, 1) / timesteps)

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def timestep(self, sigma):
        return sigma

    def sigma(self, timestep):
        return flux_time_shift(self.shift, 1.0, timestep)

    @staticmethod
    def noise_scaling(sigma, noise, latent_image):
        return sigma * noise + (1.0 - sigma) * latent_image

    @staticmethod
    def inverse_noise_scaling(sigma, latent):
        return latent / (1.0 - sigma)


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0):
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return torch.cat([sigmas, sigmas.new_zeros([1])])


def normal_scheduler(ms, steps):
    start, end = ms.timestep(ms.sigma_max), ms.timestep(ms.sigma_min)
    timesteps = torch.linspace(start, end, steps)
    sigs = [ms.sigma(timesteps[x]) for x in range(len(timesteps))]
    sigs += [0.0]
    return torch.FloatTensor(sigs)


def simple_scheduler(ms, steps):
    sigs = []
    ss = len(ms.sigmas) / steps
    for x in range(steps):
        sigs += [float(ms.sigmas[-(1 + int(x * ss))])]
    sigs += [0.0]
    return torch.FloatTensor(sigs)


def beta_scheduler(ms, steps, alpha=0.6, beta=0.6):
    import scipy.stats

    total_timesteps = len(ms.sigmas) - 1
    ts_normalized = np.linspace(0, 1, steps, endpoint=False)
    ts_beta = scipy.stats.beta.ppf(1 - ts_normalized, alpha, beta)
    ts_indices = np.rint(ts_beta * total_timesteps).astype(np.int32)
    unique_ts, indices = np.unique(ts_indices, return_index=True)
    ordered_unique_ts = unique_ts[np.argsort(indices)]
    sigs = [float(ms.sigmas[idx]) for idx in ordered_unique_ts]
    sigs.append(0.0)
    return torch.FloatTensor(sigs)


def calculate_sigmas(ms, scheduler_name, steps):
    if scheduler_name == "karras":
        return get_sigmas_karras(n=steps, sigma_min=float(ms.sigma_min), sigma_max=float(ms.sigma_max))
    if scheduler_name == "normal":
        return normal_scheduler(ms, steps)
    if scheduler_name == "simple":
        return simple_scheduler(ms, steps)
    if scheduler_name == "beta":
        return beta_scheduler(ms, steps)
    raise ValueError(f"invalid scheduler {scheduler_name}")


def sigmas_for(ms, scheduler, steps, denoise=None):
    """sample1's sigma selection incl. the denoise<1 truncation (sampling.py:966-985)."""
    if denoise is None or denoise > 0.9999:
        return calculate_sigmas(ms, scheduler, steps)
    if denoise <= 0.0:
        return torch.FloatTensor([])
    new_steps = int(steps / denoise)
    return calculate_sigmas(ms, scheduler, new_steps)[-(steps + 1):]


def prepare_noise(latent_image, seed):
    generator = torch.manual_seed(seed)
    return torch.randn(latent_image.size(), dtype=latent_image.dtype, layout=latent_image.layout,
                       generator=generator, device="cpu")


# ------------------------------------------------------------------------------------------------------
def _lcm_pad_contexts(conds):
    """CONDCrossAttn.concat (cond.py:100-126): shorter prompts are REPEATED up to the lcm length."""
    lens = [c.shape[1] for c in conds]
    if all(length == lens[0] for length in lens):
        return conds
    target = lens[0]
    for length in lens[1:]:
        target = target * length // math.gcd(target, length)
    return [c.repeat(1, target // c.shape[1], 1) if c.shape[1] < target else c for c in conds]


class CFGDenoiser:
    """CFGGuider.predict_noise -> sampling_function -> calc_cond_batch -> wrapper hook (CFG.py:86-234, cond.py:150-288).

    positive / negative: one full-area context tensor [1 or B, M, C] each, or a LIST of them (several conditioning entries per side).
    calc_cond_batch runs every entry of both sides in ONE batch, entries in reversed order and the uncond side first
    ([neg_last .. neg_0, pos_last .. pos_0] x B; cond_or_uncond == [1, .., 0, ..], cond.py:186-195), every context repeated to the lcm of the
    lengths (cond.py:100-126), and gives each side the mean of its entries' outputs (full area, multiplier 1: ksampler_util.py:106-149 of this
    snapshot; start_percent / end_percent are computed by calculate_start_end_timesteps but never consulted, so they are not mirrored).
    With one entry per side this is the [uncond x B ; cond x B] batch of ldx_unet_denoise_cfg.

    The device buffers (context, batch assembly, output) belong to the ENGINE, one set per shape, so that consecutive sampling runs replay the same
    hipGraphs.  The pair returned by __call__ are views of the engine's output buffer of that shape: valid until the next evaluation of the same
    shape on the same engine (by this or another CFGDenoiser) — the samplers consume them before they call the model again, like the reference's."""

    def __init__(self, engine, positive, negative, cfg, batch, h, w, disable_cfg1_optimization=False):
        self.engine, self.cfg = engine, float(cfg)
        dev = engine.device
        self.skip_uncond = math.isclose(self.cfg, 1.0) and not disable_cfg1_optimization
        as_list = lambda c: list(c) if isinstance(c, (list, tuple)) else [c]
        ex = lambda c: (c.to(dev, torch.float32).expand(batch, -1, -1) if c.shape[0] == 1 else c.to(dev, torch.float32))
        pos, neg = [ex(c) for c in as_list(positive)], [ex(c) for c in as_list(negative)]
        sides = ([] if self.skip_uncond else [1] * len(neg)) + [0] * len(pos)
        ctxs = ([] if self.skip_uncond else list(reversed(neg))) + list(reversed(pos))
        ctxs = _lcm_pad_contexts(ctxs)
        # Device buffers live with the ENGINE, keyed by shape, not with this object: a new sampling run (new CFGDenoiser, new latent and context
        # tensors) then presents the engine with the pointers of the previous run, and the hipGraph captured for the shape is replayed instead of
        # re-captured (measured on the reference-default multi-scale "euler": 2 captures + 2 eager evaluations of every 20 went away).
        share = hasattr(engine, "__dict__") and os.environ.get("LDX_CFG_POOL", "1") != "0"      # 0: buffers private to this object (A/B switch)
        self._pool = engine.__dict__.setdefault("_ldx_cfg_pool", {}) if share else {}
        ctx = torch.cat(ctxs).contiguous()
        key = ("ctx", tuple(ctx.shape), str(ctx.dtype), str(ctx.device))
        if key not in self._pool:
            self._pool[key] = torch.empty_like(ctx)
        self.ctx = self._pool[key]
        # the shared context buffer holds THIS object's context only while it is the owner: another CFGDenoiser of the same shape on the same engine
        # (created later, used in between) overwrites it, and __call__ then restores it from the private copy.  The owner is held through a weak
        # reference: the pool must not keep a finished run's denoiser (and its private context copy) alive.
        self._ctx_src, self._ctx_key = ctx, ("ctx_owner",) + key[1:]
        self._take_context()
        self.sides = sides                                  # cond_or_uncond of the batch
        self.n_entries = len(sides)
        self.nb = self.n_entries * batch
        self.batch = batch
        self.simple = (not self.skip_uncond) and sides == [1, 0]
        # this object's context lives in an engine-lifetime buffer that only _take_context() writes: its k|v projections may be cached across steps
        self._kw = {"ctx_cached": os.environ.get("LDX_CTX_CACHE", "1") != "0"} if hasattr(engine, "invalidate_context") else {}      # 0: recompute per call (A/B switch)

    @staticmethod
    def clear_pool(engine) -> int:
        """Drop every device buffer CFGDenoisers have parked on `engine` (one set per (batch, shape) ever sampled, kept for the engine's lifetime so that
        captured graphs are replayed across runs).  Denoisers created earlier keep working on the buffers they hold; new ones allocate afresh — and the
        engine re-captures its graph for the new pointers.  Returns the number of entries dropped.  LDX_CFG_POOL=0 never shares in the first place."""
        pool = getattr(engine, "__dict__", {}).get("_ldx_cfg_pool")
        n = len(pool) if pool else 0
        if pool:
            pool.clear()
        if n and hasattr(engine, "invalidate_context"):
            engine.invalidate_context()
        return n

    def _take_context(self):
        """Write this object's context into the engine's buffer and become its owner.  The engine caches the context's k|v projections per buffer
        (ldx_unet_context_cache: to_k / to_v of all cross-attentions are functions of the context alone — Attention.py:100-124 recomputes them every
        step); every rewrite of the buffer goes through here, so it also drops that cache."""
        import weakref
        self.ctx.copy_(self._ctx_src)
        self._pool[self._ctx_key] = weakref.ref(self)
        if hasattr(self.engine, "invalidate_context"):
            self.engine.invalidate_context()

    def _owns_context(self):
        ref = self._pool.get(self._ctx_key)
        return ref is not None and ref() is self

    def _buffers(self, shape):
        key = ("bufs", self.nb, tuple(shape), str(self.ctx.device))
        if key not in self._pool:
            b, c, h, w = shape
            dev = self.engine.device
            self._pool[key] = (torch.empty((self.nb, c, h, w), device=dev, dtype=torch.float32),
                               torch.empty((self.nb,), device=dev, dtype=torch.float32),
                               torch.empty((self.nb, c, h, w), device=dev, dtype=torch.float32),
                               torch.empty((b, c, h, w), device=dev, dtype=torch.float32))      # staging copy of x for ldx_unet_denoise_cfg
        return self._pool[key]

    def _side(self, out, side):
        """calc_cond_batch's accumulation for one side: sum of the entries' outputs in batch order / count (cond.py:262-288)."""
        b = self.batch
        idx = [i for i, s in enumerate(self.sides) if s == side]
        if len(idx) == 1:
            return out[idx[0] * b:(idx[0] + 1) * b]
        acc = torch.zeros_like(out[:b])
        for i in idx:
            acc += out[i * b:(i + 1) * b]
        return acc / (float(len(idx)) + 1e-37)

    def __call__(self, x, sigma):
        """Returns (denoised_uncond, denoised_cond), each [B,4,h,w] fp32."""
        xin, sig, out, xstage = self._buffers(x.shape)
        if not self._owns_context():
            self._take_context()
        b = self.batch
        if self.simple and hasattr(self.engine, "denoise_cfg") and x.is_cuda and x.dtype == torch.float32:
            # [uncond; cond] batch built INSIDE the engine (ldx_unet_denoise_cfg).  The engine's captured graph is tied to the pointers it was
            # captured with, and samplers hand over different tensors (dpmpp_sde's x / x2, the multi-scale steps' fresh _bilinear outputs, every new
            # sampling run): x always goes through the engine-lifetime staging tensor of its shape (one 256 KiB device copy per evaluation).
            xstage.copy_(x)
            x = xstage
            self.engine.denoise_cfg(x, float(sigma), self.ctx, out=out, **self._kw)
            return out[:b], out[b:]
        for i in range(self.n_entries):                     # the general case (cfg 1 / several entries per side): host-side batch assembly
            xin[i * b:(i + 1) * b].copy_(x)
        sig.fill_(float(sigma))
        self.engine.denoise(xin, sig, self.ctx, out=out, **self._kw)
        cond = self._side(out, 0)
        if self.skip_uncond:
            return cond, cond
        return self._side(out, 1), cond


def _step(kind, x, du, dc, cfg, c0, c1, denoised_out=None):
    L = lib.load()
    lib.check(L.ldx_sampler_step(kind, lib.ptr(x), lib.ptr(du), lib.ptr(dc), lib.ptr(denoised_out),
                                 x.numel(), float(cfg), float(c0), float(c1), lib.current_stream_ptr()),
              "ldx_sampler_step")


def _bilinear(t, size):
    L = lib.load()
    b, c, h, w = t.shape
    out = torch.empty((b, c, size[0], size[1]), device=t.device, dtype=torch.float32)
    lib.check(L.ldx_bilinear(lib.ptr(t.contiguous()), lib.ptr(out), b * c, h, w, size[0], size[1],
                             lib.current_stream_ptr()), "ldx_bilinear")
    return out


class _Multiscale:
    """The multi-scale bookkeeping shared by sample_euler / sample_dpmpp_2m_cfgpp
    (samplers.py:190-263 and 780-848)."""

    def __init__(self, shape, n_steps, enable, factor, fullres_start, fullres_end, intermittent):
        _, _, self.orig_h, self.orig_w = shape
        if enable and not (0.1 <= factor <= 1.0):
            enable = False
        if enable and (fullres_start < 0 or fullres_end < 0):
            enable = False
        self.scale_h = int(max(8, ((self.orig_h * factor) // 8) * 8)) if enable else self.orig_h
        self.scale_w = int(max(8, ((self.orig_w * factor) // 8) * 8)) if enable else self.orig_w
        self.active = enable and (self.scale_h != self.orig_h or self.scale_w != self.orig_w)
        self.n_steps, self.start, self.end, self.intermittent = n_steps, fullres_start, fullres_end, intermittent

    def fullres(self, step):
        if not self.active:
            return True
        if step < self.start or step >= self.n_steps - self.end:
            return True
        if self.intermittent:
            lo, hi = self.start, self.n_steps - self.end
            if lo <= step < hi:
                return (step - lo) % 2 == 0
        return False


@torch.no_grad()
def sample_euler(model, x, sigmas, enable_multiscale=True, multiscale_factor=0.5, multiscale_fullres_start=3,
                 multiscale_fullres_end=8, multiscale_intermittent_fullres=False, trace=None):
    """samplers.sample_euler (samplers.py:166-327) with s_churn = 0.  `model(x, sigma)` -> (uncond, cond)."""
    n_steps = len(sigmas) - 1
    ms = _Multiscale(x.shape, n_steps, enable_multiscale, multiscale_factor, multiscale_fullres_start,
                     multiscale_fullres_end, multiscale_intermittent_fullres)
    for i in range(n_steps):
        sigma_hat = sigmas[i]
        dt = sigmas[i + 1] - sigma_hat              # fp32 tensor scalars, like the reference
        if ms.fullres(i):
            du, dc = model(x, sigma_hat)
            if trace is not None:
                trace.append(tuple(x.shape[-2:]))
            _step(0, x, du, dc, model.cfg, sigma_hat, dt)
        else:
            xs = _bilinear(x, (ms.scale_h, ms.scale_w))
            if trace is not None:
                trace.append(tuple(xs.shape[-2:]))
            du, dc = model(xs, sigma_hat)
            d = torch.empty_like(xs)
            _step(2, xs, du, dc, model.cfg, 0.0, 0.0, denoised_out=d)
            d = _bilinear(d, (ms.orig_h, ms.orig_w))
            _step(0, x, d, d, 1.0, sigma_hat, dt)
    return x


@torch.no_grad()
def sample_dpmpp_2m_cfgpp(model, x, sigmas, enable_multiscale=True, multiscale_factor=0.5,
                          multiscale_fullres_start=5, multiscale_fullres_end=8,
                          multiscale_intermittent_fullres=True, trace=None):
    """samplers.sample_dpmpp_2m_cfgpp (samplers.py:754-962).  The CFG++/momentum branch never executes in
    the reference, so every step is  x = (s'/s) x - expm1(-h) * denoised."""
    n_steps = len(sigmas) - 1
    ms = _Multiscale(x.shape, n_steps, enable_multiscale, multiscale_factor, multiscale_fullres_start,
                     multiscale_fullres_end, multiscale_intermittent_fullres)
    t_steps = -torch.log(sigmas)
    sigma_steps = torch.exp(-t_steps)
    ratios = sigma_steps[1:] / sigma_steps[:-1]
    h_steps = t_steps[1:] - t_steps[:-1]
    for i in range(n_steps):
        h_expm1 = torch.expm1(-h_steps[i])
        if ms.fullres(i):
            du, dc = model(x, sigmas[i])
            if trace is not None:
                trace.append(tuple(x.shape[-2:]))
            _step(1, x, du, dc, model.cfg, ratios[i], h_expm1)
        else:
            xs = _bilinear(x, (ms.scale_h, ms.scale_w))
            if trace is not None:
                trace.append(tuple(xs.shape[-2:]))
            du, dc = model(xs, sigmas[i])
            d = torch.empty_like(xs)
            _step(2, xs, du, dc, model.cfg, 0.0, 0.0, denoised_out=d)
            d = _bilinear(d, (ms.orig_h, ms.orig_w))
            _step(1, x, d, d, 1.0, ratios[i], h_expm1)
    return x


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    """sampling_util.get_ancestral_step (sampling_util.py:128-151) on fp32 tensor scalars."""
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


@torch.no_grad()
def sample_euler_ancestral_cfgpp(model, x, sigmas, eta=1.0, s_noise=1.0, noise_sampler=None, trace=None):
    """samplers.sample_euler_ancestral_dy_cfg_pp (samplers.py:612-733) with its defaults.  The CFG++ branch is dead
    code in the reference: every step is Euler to sigma_down on the guider's ordinary CFG
    output, then  x += noise * s_noise * sigma_up  (ldx_sampler_step kind 3).  The reference's default noise sampler
    is torch.randn_like(x) on the *global* generator (sampling_util.py:154-165); to be reproducible against a CPU run
    of the reference the noise is drawn here from the global CPU generator in the same order and uploaded (256 KiB
    per image and step)."""
    n_steps = len(sigmas) - 1
    pre, k_draw = None, 0
    if noise_sampler is None:
        # the default draws do not depend on x: make them all up front in the loop's order and upload once (no blocking
        # host->device copy per step)
        n_draw = sum(1 for i in range(n_steps) if float(sigmas[i + 1]) > 0)
        pre = torch.stack([torch.randn(x.shape, dtype=torch.float32) for _ in range(n_draw)]).to(x.device) if n_draw else None
    for i in range(n_steps):
        sigma_hat = sigmas[i]
        du, dc = model(x, sigma_hat)
        if trace is not None:
            trace.append(tuple(x.shape[-2:]))
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta=eta)
        _step(0, x, du, dc, model.cfg, sigma_hat, sigma_down - sigma_hat)
        if sigmas[i + 1] > 0:
            if pre is not None:
                nz = pre[k_draw]; k_draw += 1
            else:
                nz = noise_sampler(sigmas[i], sigmas[i + 1]).to(x.device, torch.float32).contiguous()
            _step(3, x, nz, nz, 1.0, s_noise * sigma_up, 0.0)
    return x


def _dy_step_cfgpp(x, model, sigma_next, sigma_hat, current_cfg):
    """samplers.dy_sampling_step_cfg_pp (samplers.py:362-467): pixel (1,1) of every 2x2 block -> half-resolution image c,
    model at sigma_hat (the step's old sigma), a second CFG of strength current_cfg on top of the guider's (here the hook
    does see uncond_denoised), Euler update of c to sigma_next, scatter back.  The strided gather / scatter are torch
    slice copies on the device (data movement only); the arithmetic is ldx_sampler_step."""
    m, n = x.shape[2] // 2, x.shape[3] // 2
    c = x[:, :, 1:2 * m:2, 1:2 * n:2].contiguous()
    du, dc = model(c, sigma_hat)
    d = torch.empty_like(c)
    _step(2, c, du, dc, model.cfg, 0.0, 0.0, denoised_out=d)                  # guider's CFG: denoised
    _step(0, c, du, d, current_cfg, sigma_hat, sigma_next - sigma_hat)        # uncond + (denoised - uncond) * current_cfg, Euler
    x[:, :, 1:2 * m:2, 1:2 * n:2] = c
    return x


@torch.no_grad()
def sample_euler_cfgpp(model, x, sigmas, cfg_scale=7.5, cfg_min=1.0, s_extra_steps=True, trace=None):
    """samplers.sample_euler_dy_cfg_pp (samplers.py:470-609) with its defaults.  Main step = Euler on the guider's CFG
    output (the CFG++ momentum branch is dead code); after steps 2 and 3 (i // 2 == 1) the dy
    extra step runs.  `model` must evaluate both branches (disable_cfg1_optimization, samplers.py:517-520)."""
    n_steps = len(sigmas) - 1
    for i in range(n_steps):
        current_cfg = cfg_scale + (cfg_min - cfg_scale) * (i / n_steps)
        sigma_hat = sigmas[i]
        du, dc = model(x, sigma_hat)
        if trace is not None:
            trace.append(tuple(x.shape[-2:]))
        _step(0, x, du, dc, model.cfg, sigma_hat, sigmas[i + 1] - sigma_hat)
        if s_extra_steps and sigmas[i + 1] > 0 and i // 2 == 1:
            if trace is not None:
                trace.append((x.shape[-2] // 2, x.shape[-1] // 2))
            x = _dy_step_cfgpp(x, model, sigmas[i + 1], sigma_hat, current_cfg)
    return x


class BrownianIntervalNoise:
    """Stand-in for sampling_util.BrownianTreeNoiseSampler (sampling_util.py:239-292; torchsde.BrownianTree, absent offline):
    noise(sigma, sigma_next) = (W(sigma_next) - W(sigma)) / sqrt(|sigma_next - sigma|) for a Brownian motion W in sigma.
    dpmpp_sde asks, per step, for [sigma_t, sigma_s] and then for the enclosing [sigma_t, sigma_next]; the second increment
    reuses the first (W over [t, next] = W over [t, s] + an independent increment over [s, next]), so the two draws have
    the joint law a Brownian tree gives them.  The individual numbers cannot match torchsde's (its tree and seeding are
    not reproducible without the library): the sampler's arithmetic is pinned with this class injected on both sides.
    Normal draws come from a CPU generator (seed given) or from the global CPU RNG (seed None)."""

    def __init__(self, x, seed=None):
        self.shape = tuple(x.shape)
        self.gen = None if seed is None else torch.Generator().manual_seed(int(seed))
        self._t0 = self._t1 = None
        self._w = None

    def _randn(self):
        return torch.randn(self.shape, dtype=torch.float32, generator=self.gen)

    def __call__(self, sigma, sigma_next):
        t0, t1 = float(sigma), float(sigma_next)
        if self._t0 is not None and t0 == self._t0 and abs(t1 - t0) > abs(self._t1 - t0) and (t1 - t0) * (self._t1 - t0) > 0:
            w = self._w + self._randn() * math.sqrt(abs(t1 - self._t1))      # extend [t0, s] to [t0, t1]
        else:
            w = self._randn() * math.sqrt(abs(t1 - t0))
        self._t0, self._t1, self._w = t0, t1, w
        return w / math.sqrt(abs(t1 - t0))


@torch.no_grad()
def sample_dpmpp_sde_cfgpp(model, x, sigmas, eta=1.0, s_noise=1.0, noise_sampler=None, r=0.5, seed=None, enable_multiscale=True,
                           multiscale_factor=0.5, multiscale_fullres_start=5, multiscale_fullres_end=8,
                           multiscale_intermittent_fullres=False, trace=None):
    """samplers.sample_dpmpp_sde_cfgpp (samplers.py:965-1254) with its defaults (r = 1/2).  As in the other cfgpp samplers the
    CFG++ momentum branch is dead code: the sampler calls its own post-cfg hook with uncond_denoised = None (:1137-1139,
    :1209-1211), which resets old_uncond_denoised and returns the guider's CFG output, so cfg_denoised == denoised.  What
    runs is DPM-Solver++(SDE): per step two model evaluations (at sigma_i and at the midpoint sigma_fn(s) in log-sigma),
    each followed by an ancestral split (sampling_util.get_ancestral_step) and a noise injection; the last step
    (sigma_next == 0) is one Euler step.  Low-resolution steps evaluate BOTH model calls at the reduced size (:1186)."""
    assert r == 0.5, "the reference never changes r; (1 - 1/(2r)) = 0 is assumed"
    n_steps = len(sigmas) - 1
    if n_steps < 1:
        return x
    ms = _Multiscale(x.shape, n_steps, enable_multiscale, multiscale_factor, multiscale_fullres_start,
                     multiscale_fullres_end, multiscale_intermittent_fullres)
    sigma_fn = lambda t: torch.exp(-t)       # noqa: E731
    t_fn = lambda sg: -torch.log(sg)         # noqa: E731
    # Default noise (BrownianIntervalNoise): the draws do not depend on x, so all of them are made up front, in the order the
    # per-call sampler would make them (two per step with sigma_next > 0), and uploaded once; the Brownian combination
    # (W[t,next] = W[t,s] + increment over [s,next]) becomes two noise-injection steps with host-computed coefficients.
    # No host<->device synchronisation is left in the loop (the per-call path costs one blocking upload per evaluation).
    pre = None
    if noise_sampler is None:
        ref = BrownianIntervalNoise(x, seed)
        n_draw = 2 * sum(1 for i in range(n_steps) if float(sigmas[i + 1]) != 0.0)
        pre = torch.stack([ref._randn() for _ in range(n_draw)]).to(x.device) if n_draw else None
    k_draw = 0

    def denoise(xx, sigma, full):
        """guider CFG output at the step's resolution, returned at full resolution"""
        xp = xx if full else _bilinear(xx, (ms.scale_h, ms.scale_w))
        if trace is not None:
            trace.append(tuple(xp.shape[-2:]))
        du, dc = model(xp, sigma)
        d = torch.empty_like(xp)
        _step(2, xp, du, dc, model.cfg, 0.0, 0.0, denoised_out=d)
        return d if full else _bilinear(d, (ms.orig_h, ms.orig_w))

    for i in range(n_steps):
        full = ms.fullres(i)
        d1 = denoise(x, sigmas[i], full)
        if sigmas[i + 1] == 0:
            _step(0, x, d1, d1, 1.0, sigmas[i], sigmas[i + 1] - sigmas[i])                     # Euler on to_d (:1156-1159)
            continue
        t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
        s = t + (t_next - t) * r
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(s), eta)
        s_ = t_fn(sd)
        x2 = x.clone()
        _step(1, x2, d1, d1, 1.0, sigma_fn(s_) / sigma_fn(t), torch.expm1(t - s_))            # (sigma(s_)/sigma(t)) x - expm1(t - s_) d
        if pre is not None:
            z1, z2 = pre[k_draw], pre[k_draw + 1]
            k_draw += 2
            _step(3, x2, z1, z1, 1.0, s_noise * su, 0.0)                                       # W[t,s] / sqrt|s - t| = z1
        else:
            nz = noise_sampler(sigma_fn(t), sigma_fn(s)).to(x.device, torch.float32).contiguous()
            _step(3, x2, nz, nz, 1.0, s_noise * su, 0.0)
        d2 = denoise(x2, sigma_fn(s), full)
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(t_next), eta)
        t_next_ = t_fn(sd)
        _step(1, x, d2, d2, 1.0, sigma_fn(t_next_) / sigma_fn(t), torch.expm1(t - t_next_))
        if pre is not None:
            st, ss, sn = float(sigma_fn(t)), float(sigma_fn(s)), float(sigma_fn(t_next))
            total = abs(sn - st)
            _step(3, x, z1, z1, 1.0, float(s_noise * su) * math.sqrt(abs(ss - st) / total), 0.0)
            _step(3, x, z2, z2, 1.0, float(s_noise * su) * math.sqrt(abs(sn - ss) / total), 0.0)
        else:
            nz = noise_sampler(sigma_fn(t), sigma_fn(t_next)).to(x.device, torch.float32).contiguous()
            _step(3, x, nz, nz, 1.0, s_noise * su, 0.0)
    return x


_MULTISCALE_WHITELIST = ("dpmpp_sde_cfgpp", "sample_euler_ancestral", "sample_euler", "sample_dpmpp_2m_cfgpp")


def _resolve_sampler(sampler_name):
    """sampling.ksampler (sampling.py:500-534): only four names are recognised, the rest fall back to Euler."""
    if sampler_name == "dpmpp_2m_cfgpp":
        return sample_dpmpp_2m_cfgpp, True
    if sampler_name == "euler_ancestral_cfgpp":
        return sample_euler_ancestral_cfgpp, True
    if sampler_name == "euler_cfgpp":
        return sample_euler_cfgpp, True
    if sampler_name == "dpmpp_sde_cfgpp":
        return sample_dpmpp_sde_cfgpp, True
    return sample_euler, False


class KSampler:
    """KSampler.sample -> common_ksampler -> sample1 -> CFGGuider.sample (sampling.py:773-1233, CFG.py:164-357)
    for txt2img / img2img latents; positive / negative are one full-area context each or lists of them (CFGDenoiser)."""

    def __init__(self, engine):
        self.engine = engine
        self.model_sampling = ModelSamplingDiscrete()

    def sample(self, seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, denoise=1.0,
               enable_multiscale=True, multiscale_factor=0.5, multiscale_fullres_start=3,
               multiscale_fullres_end=8, multiscale_intermittent_fullres=False, noise=None, trace=None):
        ms = self.model_sampling
        denoise = denoise or 1.0                                   # sampling.py:875 quirk (A-12)
        latent_image = latent_image.float()
        if noise is None:
            noise = prepare_noise(latent_image, seed)
        sigmas = sigmas_for(ms, scheduler, steps, denoise)
        fn, disable_cfg1 = _resolve_sampler(sampler_name)
        extra = {}
        if sampler_name in _MULTISCALE_WHITELIST:                  # whitelist mismatch quirk (A-3)
            extra = dict(enable_multiscale=enable_multiscale, multiscale_factor=multiscale_factor,
                         multiscale_fullres_start=multiscale_fullres_start,
                         multiscale_fullres_end=multiscale_fullres_end,
                         multiscale_intermittent_fullres=multiscale_intermittent_fullres)
        dev = self.engine.device
        b, _, h, w = latent_image.shape
        # CFGGuider.inner_sample (CFG.py:266-269): an all-zero latent is not shifted
        if torch.count_nonzero(latent_image) > 0:
            latent_image = latent_image * 0.18215                  # process_latent_in (Latent.py:19-39)
        # KSAMPLER.sample noise scaling (sampling.py:58-83, 410-422)
        max_sigma, s0 = float(ms.sigma_max), float(sigmas[0])
        if math.isclose(max_sigma, s0, rel_tol=1e-05) or s0 > max_sigma:
            x = noise * torch.sqrt(1.0 + sigmas[0] ** 2.0)
        else:
            x = noise * sigmas[0]
        x = x.to(dev) + latent_image.to(dev)                       # the latent may already live on the device (HiresFix chain)
        model = CFGDenoiser(self.engine, positive, negative, cfg, b, h, w, disable_cfg1_optimization=disable_cfg1)
        x = fn(model, x, sigmas, trace=trace, **extra)
        return x / 0.18215                                         # process_latent_out (CFG.py:294)


# ------------------------------------------------------------------------------------------------------
# Flux sampling path: KSampler.sample(flux=True), pipeline.py:237-262
FLUX_LATENT_SCALE, FLUX_LATENT_SHIFT = 0.3611, 0.1159          # Latent.Flux1 (Latent.py:114-161)


class FluxCFGDenoiser:
    """CFGGuider.predict_noise for Flux2 (CFG.py:86-234; Flux2.extra_conds, Flux.py:800-815): conditioning = T5 context,
    pooled CLIP vector (y) and the guidance scalar; batch order [uncond x B ; cond x B]."""

    def __init__(self, engine, positive, negative, cfg, guidance, batch, disable_cfg1_optimization=False):
        self.engine, self.cfg = engine, float(cfg)
        dev = engine.device
        self.skip_uncond = math.isclose(self.cfg, 1.0) and not disable_cfg1_optimization
        (pc, py), (nc, ny) = positive, negative
        f = lambda t: t.to(dev, torch.float32)          # noqa: E731
        pc, py = f(pc).expand(batch, -1, -1), f(py).expand(batch, -1)
        if self.skip_uncond:
            self.ctx, self.y, self.nb = pc.contiguous(), py.contiguous(), batch
        else:
            nc, ny = f(nc).expand(batch, -1, -1), f(ny).expand(batch, -1)
            nc, pc = _lcm_pad_contexts([nc, pc])
            self.ctx, self.y, self.nb = torch.cat([nc, pc]).contiguous(), torch.cat([ny, py]).contiguous(), 2 * batch
        self.guidance = torch.full((self.nb,), float(guidance), device=dev, dtype=torch.float32)
        self.batch = batch

    def __call__(self, x, sigma):
        b = self.batch
        xin = x if self.skip_uncond else torch.cat([x, x])
        sig = torch.full((self.nb,), float(sigma), device=x.device, dtype=torch.float32)
        out = self.engine.denoise(xin.contiguous(), sig, self.ctx, self.y, self.guidance)
        if self.skip_uncond:
            return out, out
        return out[:b], out[b:]


class FluxKSampler:
    """KSampler.sample(flux=True) -> CFGGuider.sample -> KSAMPLER.sample (sampling.py:444-498, 773-1233; CFG.py:236-357) for
    one positive and one negative (ctx, pooled) pair."""

    def __init__(self, engine, shift=1.15):
        self.engine = engine
        self.model_sampling = ModelSamplingFlux(shift)

    def sample(self, seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, guidance=3.0, denoise=1.0,
               enable_multiscale=True, multiscale_factor=0.5, multiscale_fullres_start=3, multiscale_fullres_end=8,
               multiscale_intermittent_fullres=False, noise=None, trace=None):
        ms = self.model_sampling
        denoise = denoise or 1.0
        latent_image = latent_image.float()
        if noise is None:
            noise = prepare_noise(latent_image, seed)
        sigmas = sigmas_for(ms, scheduler, steps, denoise)
        fn, disable_cfg1 = _resolve_sampler(sampler_name)
        extra = {}
        if sampler_name in _MULTISCALE_WHITELIST:
            extra = dict(enable_multiscale=enable_multiscale, multiscale_factor=multiscale_factor,
                         multiscale_fullres_start=multiscale_fullres_start, multiscale_fullres_end=multiscale_fullres_end,
                         multiscale_intermittent_fullres=multiscale_intermittent_fullres)
        if torch.count_nonzero(latent_image) > 0:                            # CFG.py:266-269
            latent_image = (latent_image - FLUX_LATENT_SHIFT) * FLUX_LATENT_SCALE
        x = ms.noise_scaling(sigmas[0], noise, latent_image).to(self.engine.device)
        model = FluxCFGDenoiser(self.engine, positive, negative, cfg, guidance, x.shape[0], disable_cfg1_optimization=disable_cfg1)
        x = fn(model, x, sigmas, trace=trace, **extra)
        x = ms.inverse_noise_scaling(sigmas[-1], x)
        return x / FLUX_LATENT_SCALE + FLUX_LATENT_SHIFT                    # process_latent_out

