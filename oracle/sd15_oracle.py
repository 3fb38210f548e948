"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path (lightdiffusion-next_amd/).

CPU restatement (plain PyTorch fp32 functional ops) of the reference's denoising hot path for SD1.5, each
function citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

Parity status: the reference (Aatricks/LightDiffusion-Next) has NO tests, golden vectors or KATs
 — "parity unpinned" by the reference's own suite.  This oracle is therefore pinned
against outputs of the reference itself, captured in the build container by importing /root/reference
(oracle/ref_capture.py -> tests/golden/*.npz; checked in tests/test_oracle_vs_golden.py).

Numerics: the reference CPU path stores UNet weights in fp16 and computes in fp32 via cast.manual_cast
(src/Device/Device.py:980-1012, src/cond/cast.py:44-78, 479-525).  Upcasting the fp16 weights to fp32 once
and running plain fp32 ops is bit-identical to that path (probed); this file does the latter.
"""
import math

import numpy as np

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------
# schedule / model sampling  (src/sample/sampling.py:221-356, src/sample/sampling_util.py:18-39)
def make_sigmas():
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2   # sampling_util.py:18-39
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)                                        # sampling.py:268-270
    sigmas = ((1 - alphas_cumprod) / alphas_cumprod) ** 0.5                                   # sampling.py:276
    return sigmas.float(), sigmas.log().float()                                               # sampling.py:285-289


SIGMAS, LOG_SIGMAS = make_sigmas()


def timestep(sigma):
    """ModelSamplingDiscrete.timestep (sampling.py:309-320): nearest log-sigma table index."""
    log_sigma = sigma.log()
    dists = log_sigma - LOG_SIGMAS[:, None]
    return dists.abs().argmin(dim=0).view(sigma.shape)


def sigma_of(t):
    """ModelSamplingDiscrete.sigma (sampling.py:322-339)."""
    t = torch.clamp(t.float(), min=0, max=(len(SIGMAS) - 1))
    low_idx, high_idx, w = t.floor().long(), t.ceil().long(), t.frac()
    log_sigma = (1 - w) * LOG_SIGMAS[low_idx] + w * LOG_SIGMAS[high_idx]
    return log_sigma.exp()


def timestep_embedding(timesteps, dim, max_period=10000):
    """sampling_util.py:56-76."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


# ---------------------------------------------------------------------------------------------------
class W:
    """fp32 view of a (possibly fp16) state dict with prefix navigation."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name):
        return self.sd[self.prefix + name].float()

    def has(self, name):
        return (self.prefix + name) in self.sd

    def sub(self, p):
        return W(self.sd, self.prefix + p)


def group_norm(x, w, name, eps):
    return F.group_norm(x, 32, w(name + ".weight"), w(name + ".bias"), eps)           # cast.py:241-243


def resblock(w, x, emb):
    """ResBlock1._forward (src/AutoEncoders/ResBlock.py:315-335); GroupNorm eps 1e-5 (ResBlock.py:252,280)."""
    h = F.silu(group_norm(x, w, "in_layers.0", 1e-5))
    h = F.conv2d(h, w("in_layers.2.weight"), w("in_layers.2.bias"), padding=1)
    emb_out = F.linear(F.silu(emb), w("emb_layers.1.weight"), w("emb_layers.1.bias"))
    h = h + emb_out[..., None, None]
    h = F.silu(group_norm(h, w, "out_layers.0", 1e-5))
    h = F.conv2d(h, w("out_layers.3.weight"), w("out_layers.3.bias"), padding=1)
    if w.has("skip_connection.weight"):
        x = F.conv2d(x, w("skip_connection.weight"), w("skip_connection.bias"))
    return x + h


def attention(q, k, v, heads):
    """attention_pytorch (src/Attention/AttentionMethods.py:107-150): [B,N,H*D] -> SDPA -> [B,N,H*D]."""
    b, _, inner = q.shape
    d = inner // heads
    q, k, v = (t.view(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    out = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
    return out.transpose(1, 2).reshape(b, -1, heads * d)


def cross_attention(w, x, context, heads):
    """CrossAttention.forward (src/Attention/Attention.py:100-124): q/k/v without bias, out-proj with bias."""
    context = x if context is None else context
    q = F.linear(x, w("to_q.weight"))
    k = F.linear(context, w("to_k.weight"))
    v = F.linear(context, w("to_v.weight"))
    out = attention(q, k, v, heads)
    return F.linear(out, w("to_out.0.weight"), w("to_out.0.bias"))


def basic_transformer_block(w, x, context, heads):
    """BasicTransformerBlock._forward (src/NeuralNetwork/transformer.py:186-245); LayerNorm eps 1e-5;
    FeedForward with GEGLU, erf-GELU (transformer.py:19-70, src/cond/Activation.py:6-31)."""
    c = x.shape[-1]
    n = F.layer_norm(x, (c,), w("norm1.weight"), w("norm1.bias"))
    x = x + cross_attention(w.sub("attn1."), n, None, heads)
    n = F.layer_norm(x, (c,), w("norm2.weight"), w("norm2.bias"))
    x = x + cross_attention(w.sub("attn2."), n, context, heads)
    n = F.layer_norm(x, (c,), w("norm3.weight"), w("norm3.bias"))
    a, gate = F.linear(n, w("ff.net.0.proj.weight"), w("ff.net.0.proj.bias")).chunk(2, dim=-1)
    ff = F.linear(a * F.gelu(gate), w("ff.net.2.weight"), w("ff.net.2.bias"))
    return ff + x


def spatial_transformer(w, x, context, heads, depth):
    """SpatialTransformer.forward (transformer.py:342-377), use_linear=False; GroupNorm eps 1e-6 (:286-293)."""
    b, c, h, wd = x.shape
    x_in = x
    x = group_norm(x, w, "norm", 1e-6)
    x = F.conv2d(x, w("proj_in.weight"), w("proj_in.bias"))
    x = x.permute(0, 2, 3, 1).reshape(b, h * wd, c)
    for d in range(depth):
        x = basic_transformer_block(w.sub(f"transformer_blocks.{d}."), x, context, heads)
    x = x.reshape(b, h, wd, c).permute(0, 3, 1, 2)
    x = F.conv2d(x, w("proj_out.weight"), w("proj_out.bias"))
    return x + x_in


def unet_forward(sd, cfg, x, timesteps, context):
    """UNetModel1.forward (src/NeuralNetwork/unet.py:679-770) for the SD1.5 family; structure walk as in
    UNetModel1.__init__ (unet.py:344-677).  x fp32 NCHW, timesteps [B] (integer-valued), context [B,M,ctx]."""
    w = W(sd)
    mc, heads = cfg.model_channels, cfg.num_heads
    t_emb = timestep_embedding(timesteps, mc)
    emb = F.linear(F.silu(F.linear(t_emb, w("time_embed.0.weight"), w("time_embed.0.bias"))),
                   w("time_embed.2.weight"), w("time_embed.2.bias"))
    td, tdo = list(cfg.transformer_depth), list(cfg.transformer_depth_output)
    nl = len(cfg.channel_mult)
    hs = []
    h = F.conv2d(x, w("input_blocks.0.0.weight"), w("input_blocks.0.0.bias"), padding=1)
    hs.append(h)
    ib = 1
    for level in range(nl):
        for _ in range(cfg.num_res_blocks[level]):
            h = resblock(w.sub(f"input_blocks.{ib}.0."), h, emb)
            depth = td.pop(0)
            if depth > 0:
                h = spatial_transformer(w.sub(f"input_blocks.{ib}.1."), h, context, heads, depth)
            hs.append(h)
            ib += 1
        if level != nl - 1:
            h = F.conv2d(h, w(f"input_blocks.{ib}.0.op.weight"), w(f"input_blocks.{ib}.0.op.bias"), stride=2, padding=1)
            hs.append(h)                                                                  # Downsample1 ResBlock.py:141-194
            ib += 1
    h = resblock(w.sub("middle_block.0."), h, emb)
    if cfg.transformer_depth_middle >= 0:
        h = spatial_transformer(w.sub("middle_block.1."), h, context, heads, cfg.transformer_depth_middle)
        h = resblock(w.sub("middle_block.2."), h, emb)
    ob = 0
    for level in reversed(range(nl)):
        for i in range(cfg.num_res_blocks[level] + 1):
            h = torch.cat([h, hs.pop()], dim=1)                                           # unet.py:750
            output_shape = hs[-1].shape if hs else None                                   # unet.py:754
            h = resblock(w.sub(f"output_blocks.{ob}.0."), h, emb)
            sub = 1
            depth = tdo.pop()
            if depth > 0:
                h = spatial_transformer(w.sub(f"output_blocks.{ob}.1."), h, context, heads, depth)
                sub += 1
            if level and i == cfg.num_res_blocks[level]:                                  # Upsample1 ResBlock.py:75-138
                shape = [h.shape[2] * 2, h.shape[3] * 2]
                if output_shape is not None:
                    shape = [output_shape[2], output_shape[3]]
                h = F.interpolate(h, size=shape, mode="nearest")
                h = F.conv2d(h, w(f"output_blocks.{ob}.{sub}.conv.weight"), w(f"output_blocks.{ob}.{sub}.conv.bias"), padding=1)
            ob += 1
    h = F.silu(group_norm(h, w, "out.0", 1e-5))                                           # unet.py:663-677
    return F.conv2d(h, w("out.2.weight"), w("out.2.bias"), padding=1)


def apply_model(sd, cfg, x, sigma, context, c_concat=None):
    """BaseModel.apply_model (src/Model/ModelBase.py:72-133) with EPS (sampling.py:26-56):
    xc = x / sqrt(sigma^2 + 1); t = timestep(sigma).float(); denoised = x - out * sigma.
    c_concat (ModelBase.py:100-101, inpainting UNets): appended UNSCALED behind the scaled x; the denoised latent still is x - out * sigma."""
    s = sigma.view(sigma.shape[:1] + (1,) * (x.ndim - 1))
    xc = x / (s ** 2 + 1.0) ** 0.5
    if c_concat is not None:
        xc = torch.cat((xc, c_concat), dim=1)
    t = timestep(sigma).float()
    out = unet_forward(sd, cfg, xc.float(), t, context.float()).float()
    return x - out * s


# ---------------------------------------------------------------------------------------------------
# scheduler loop  (src/sample/*, src/cond/cond.py)
def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0):
    """sampling_util.py:106-125."""
    ramp = torch.linspace(0, 1, n)
    min_inv_rho, max_inv_rho = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return torch.cat([sigmas, sigmas.new_zeros([1])])


def calculate_sigmas(scheduler_name, steps, SIGMAS=None):
    """ksampler_util.py:152-271.  SIGMAS: the model-sampling table (default: SD1.5's 1000 discrete sigmas; Flux passes
    its 10000-entry shifted table, for which only "simple" and "beta" work in the reference — "normal"/"karras" need
    sigma_min, which ModelSamplingFlux lacks)."""
    flux = SIGMAS is not None
    SIGMAS = globals()["SIGMAS"] if SIGMAS is None else SIGMAS
    if flux and scheduler_name in ("karras", "normal"):
        raise AttributeError("'ModelSampling' object has no attribute 'sigma_min'")
    if scheduler_name == "karras":
        return get_sigmas_karras(steps, float(SIGMAS[0]), float(SIGMAS[-1]))
    if scheduler_name == "normal":
        start, end = timestep(SIGMAS[-1]), timestep(SIGMAS[0])
        ts = torch.linspace(start, end, steps)
        return torch.FloatTensor([sigma_of(ts[i]) for i in range(len(ts))] + [0.0])
    if scheduler_name == "simple":
        ss = len(SIGMAS) / steps
        return torch.FloatTensor([float(SIGMAS[-(1 + int(x * ss))]) for x in range(steps)] + [0.0])
    if scheduler_name == "beta":
        import numpy as np
        import scipy.stats
        total = len(SIGMAS) - 1
        ts = 1 - np.linspace(0, 1, steps, endpoint=False)
        idx = np.rint(scipy.stats.beta.ppf(ts, 0.6, 0.6) * total).astype(np.int32)
        uniq, first = np.unique(idx, return_index=True)
        ordered = uniq[np.argsort(first)]
        return torch.FloatTensor([float(SIGMAS[i]) for i in ordered] + [0.0])
    raise ValueError(scheduler_name)


def sigmas_for(scheduler, steps, denoise, SIGMAS=None):
    """sampling.py:966-985 / KSampler.set_steps :665-676."""
    if denoise is None or denoise > 0.9999:
        return calculate_sigmas(scheduler, steps, SIGMAS)
    new_steps = int(steps / denoise)
    return calculate_sigmas(scheduler, new_steps, SIGMAS)[-(steps + 1):]


def lcm_pad(conds):
    """CONDCrossAttn.concat (cond.py:100-126)."""
    lens = [c.shape[1] for c in conds]
    if all(l == lens[0] for l in lens):
        return conds
    target = lens[0]
    for l in lens[1:]:
        target = target * l // math.gcd(target, l)
    return [c.repeat(1, target // c.shape[1], 1) if c.shape[1] < target else c for c in conds]


def _side_mean(outs):
    """calc_cond_batch's accumulation for full-area entries (cond.py:262-288): out += output * 1, count += 1, out /= count
    (count starts at 1e-37, which fp32 absorbs): the plain mean of the side's entries, summed in BATCH order."""
    acc = torch.zeros_like(outs[0])
    cnt = torch.ones_like(outs[0]) * 1e-37
    for o in outs:
        acc = acc + o
        cnt = cnt + 1.0
    return acc / cnt


def cfg_denoise_multi(denoiser, x, sigma, positives, negatives, cfg, disable_cfg1_optimization=False):
    """calc_cond_batch with SEVERAL conditioning entries per side (cond.py:150-288; get_area_and_mult of this snapshot gives every entry the
    full area and multiplier 1, ksampler_util.py:106-149): to_run = positives then negatives, batched in REVERSED order (cond.py:186-194:
    [neg_last .. neg_0, pos_last .. pos_0]), contexts padded to the lcm of all lengths (cond_cat), each side = the mean of its entries."""
    b = x.shape[0]
    ex = lambda c: c.expand(b, -1, -1) if c.shape[0] == 1 else c
    skip_uncond = math.isclose(cfg, 1.0) and not disable_cfg1_optimization
    entries = ([] if skip_uncond else [(ex(c), 1) for c in reversed(negatives)]) + [(ex(c), 0) for c in reversed(positives)]
    ctxs = lcm_pad([e[0] for e in entries])
    n = len(entries)
    out = denoiser(torch.cat([x] * n), sigma * x.new_ones([n * b]), torch.cat(ctxs)).chunk(n)
    cond = _side_mean([o for o, e in zip(out, entries) if e[1] == 0])
    if skip_uncond or math.isclose(cfg, 1.0):
        return cond
    uncond = _side_mean([o for o, e in zip(out, entries) if e[1] == 1])
    return torch.lerp(uncond, cond, cfg)


def cfg_denoise(denoiser, x, sigma, positive, negative, cfg, disable_cfg1_optimization=False):
    """sampling_function + calc_cond_batch + cfg_function (CFG.py:6-161, cond.py:150-288): one batched call
    in [uncond; cond] order; torch.lerp for the combine.  positive / negative: one context tensor, or a list of them (several entries per side)."""
    if isinstance(positive, (list, tuple)) or isinstance(negative, (list, tuple)):
        as_list = lambda c: list(c) if isinstance(c, (list, tuple)) else [c]
        return cfg_denoise_multi(denoiser, x, sigma, as_list(positive), as_list(negative), cfg, disable_cfg1_optimization)
    b = x.shape[0]
    pos = positive.expand(b, -1, -1) if positive.shape[0] == 1 else positive
    if math.isclose(cfg, 1.0) and not disable_cfg1_optimization:
        return denoiser(x, sigma * x.new_ones([b]), pos)
    neg = negative.expand(b, -1, -1) if negative.shape[0] == 1 else negative
    neg, pos = lcm_pad([neg, pos])
    out = denoiser(torch.cat([x, x]), sigma * x.new_ones([2 * b]), torch.cat([neg, pos]))
    uncond, cond = out.chunk(2)
    if math.isclose(cfg, 1.0):
        return cond
    return torch.lerp(uncond, cond, cfg)


class Multiscale:
    """samplers.py:190-263."""

    def __init__(self, shape, n_steps, enable, factor, start, end, intermittent):
        self.oh, self.ow = shape[-2:]
        if enable and not (0.1 <= factor <= 1.0):
            enable = False
        if enable and (start < 0 or end < 0):
            enable = False
        self.sh = int(max(8, ((self.oh * factor) // 8) * 8)) if enable else self.oh
        self.sw = int(max(8, ((self.ow * factor) // 8) * 8)) if enable else self.ow
        self.active = enable and (self.sh != self.oh or self.sw != self.ow)
        self.n, self.start, self.end, self.intermittent = n_steps, start, end, intermittent

    def fullres(self, i):
        if not self.active:
            return True
        if i < self.start or i >= self.n - self.end:
            return True
        if self.intermittent and self.start <= i < self.n - self.end:
            return (i - self.start) % 2 == 0
        return False

    def down(self, t):
        return F.interpolate(t, size=(self.sh, self.sw), mode="bilinear", align_corners=False)

    def up(self, t):
        return F.interpolate(t, size=(self.oh, self.ow), mode="bilinear", align_corners=False)


def sample_euler(model, x, sigmas, enable_multiscale=True, multiscale_factor=0.5, multiscale_fullres_start=3,
                 multiscale_fullres_end=8, multiscale_intermittent_fullres=False, trace=None):
    """samplers.sample_euler (samplers.py:166-327), s_churn = 0.  model(x, sigma) -> CFG-combined denoised."""
    n = len(sigmas) - 1
    ms = Multiscale(x.shape, n, enable_multiscale, multiscale_factor, multiscale_fullres_start,
                    multiscale_fullres_end, multiscale_intermittent_fullres)
    for i in range(n):
        sigma_hat = sigmas[i]
        full = ms.fullres(i)
        xp = x if full else ms.down(x)
        if trace is not None:
            trace.append(tuple(xp.shape[-2:]))
        denoised = model(xp, sigma_hat)
        if not full:
            denoised = ms.up(denoised)
        x = x + ((x - denoised) / sigma_hat) * (sigmas[i + 1] - sigma_hat)                # util.to_d util.py:26-37
    return x


def sample_dpmpp_2m_cfgpp(model, x, sigmas, enable_multiscale=True, multiscale_factor=0.5,
                          multiscale_fullres_start=5, multiscale_fullres_end=8,
                          multiscale_intermittent_fullres=True, trace=None):
    """samplers.sample_dpmpp_2m_cfgpp (samplers.py:754-962).  old_uncond_denoised is reset to None by the manual
    post-CFG hook call (samplers.py:910-912) so the first-order branch always runs."""
    n = len(sigmas) - 1
    ms = Multiscale(x.shape, n, enable_multiscale, multiscale_factor, multiscale_fullres_start,
                    multiscale_fullres_end, multiscale_intermittent_fullres)
    t_steps = -torch.log(sigmas)
    sigma_steps = torch.exp(-t_steps)
    ratios = sigma_steps[1:] / sigma_steps[:-1]
    h_steps = t_steps[1:] - t_steps[:-1]
    for i in range(n):
        full = ms.fullres(i)
        xp = x if full else ms.down(x)
        if trace is not None:
            trace.append(tuple(xp.shape[-2:]))
        denoised = model(xp, sigmas[i])
        if not full:
            denoised = ms.up(denoised)
        x = ratios[i] * x - torch.expm1(-h_steps[i]) * denoised
    return x


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    """sampling_util.py:128-151."""
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def sample_euler_ancestral_cfgpp(model, x, sigmas, eta=1.0, s_noise=1.0, trace=None):
    """samplers.sample_euler_ancestral_dy_cfg_pp (samplers.py:612-733) with its defaults (s_gamma_* = 0 so
    sigma_hat = sigma_i).  The hand-made post-CFG hook call makes uncond_denoised *be* denoised (SURVEY Appendix
    A-2), so cfg_denoised == denoised and this is Euler-ancestral on the guider's ordinary CFG output.  Noise comes
    from default_noise_sampler (sampling_util.py:154-165): torch.randn_like(x) from the *global* RNG, i.e. the
    stream prepare_noise seeded, continued after the initial noise draw."""
    n = len(sigmas) - 1
    for i in range(n):
        sigma_hat = sigmas[i]
        if trace is not None:
            trace.append(tuple(x.shape[-2:]))
        denoised = model(x, sigma_hat)
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta=eta)
        x = x + ((x - denoised) / sigma_hat) * (sigma_down - sigma_hat)
        if sigmas[i + 1] > 0:
            x = x + torch.randn_like(x) * s_noise * sigma_up
    return x


def dy_sampling_step_cfg_pp(x, model_uc, cfg, sigma_next, sigma_hat, current_cfg):
    """samplers.dy_sampling_step_cfg_pp (samplers.py:362-467): the (1,1) pixel of every 2x2 block forms a half-resolution
    image c; the model is evaluated on c at sigma_hat (the step's *old* sigma, although x is already at sigma_next); here
    the hook really sees uncond_denoised, so a second CFG of strength current_cfg is applied on top of the guider's:
    cfg_denoised = uncond + (denoised - uncond) * current_cfg; Euler update of c to sigma_next; scatter back."""
    b, ch, hh, ww = x.shape
    m, n = hh // 2, ww // 2
    c = x[:, :, 1:2 * m:2, 1:2 * n:2].clone()
    uncond, cond = model_uc(c, sigma_hat)
    denoised = torch.lerp(uncond, cond, cfg) if not math.isclose(cfg, 1.0) else cond
    cfg_denoised = uncond + (denoised - uncond) * current_cfg
    c = c + ((c - cfg_denoised) / sigma_hat) * (sigma_next - sigma_hat)
    x = x.clone()
    x[:, :, 1:2 * m:2, 1:2 * n:2] = c
    return x


def sample_euler_cfgpp(model_uc, x, sigmas, cfg, cfg_scale=7.5, cfg_min=1.0, trace=None):
    """samplers.sample_euler_dy_cfg_pp (samplers.py:470-609) with its defaults (s_churn 0, s_gamma_* 0 => sigma_hat =
    sigma_i).  The main step is plain Euler on the guider's output (CFG++ branch dead); after steps
    i with i // 2 == 1 (and sigma_{i+1} > 0) comes the dy extra step.  model_uc(x, sigma) -> (uncond, cond): the sampler
    sets disable_cfg1_optimization, so both are always evaluated."""
    n_steps = len(sigmas) - 1
    for i in range(n_steps):
        current_cfg = cfg_scale + (cfg_min - cfg_scale) * (i / n_steps)
        sigma_hat = sigmas[i]
        if trace is not None:
            trace.append(tuple(x.shape[-2:]))
        uncond, cond = model_uc(x, sigma_hat)
        denoised = torch.lerp(uncond, cond, cfg) if not math.isclose(cfg, 1.0) else cond
        x = x + ((x - denoised) / sigma_hat) * (sigmas[i + 1] - sigma_hat)
        if sigmas[i + 1] > 0 and i // 2 == 1:
            if trace is not None:
                trace.append((x.shape[-2] // 2, x.shape[-1] // 2))
            x = dy_sampling_step_cfg_pp(x, model_uc, cfg, sigmas[i + 1], sigma_hat, current_cfg)
    return x


class BrownianIntervalNoise:
    """The build's stand-in for BrownianTreeNoiseSampler (sampling_util.py:239-292; torchsde is absent offline, so the
    tree's own numbers are out of reach — "parity unpinned" for the noise values; the sampler arithmetic is pinned with
    this class injected into the reference, oracle/ref_capture_sde.py).  (W(s1) - W(s0)) / sqrt|s1 - s0| of a Brownian
    motion in sigma; a query that extends the previous interval from the same start reuses its increment."""

    def __init__(self, x, seed=None):
        self.shape = tuple(x.shape)
        self.gen = None if seed is None else torch.Generator().manual_seed(int(seed))
        self.t0 = self.t1 = self.w = None

    def __call__(self, sigma, sigma_next):
        t0, t1 = float(sigma), float(sigma_next)
        z = torch.randn(self.shape, dtype=torch.float32, generator=self.gen)
        if self.t0 is not None and t0 == self.t0 and abs(t1 - t0) > abs(self.t1 - t0) and (t1 - t0) * (self.t1 - t0) > 0:
            w = self.w + z * math.sqrt(abs(t1 - self.t1))
        else:
            w = z * math.sqrt(abs(t1 - t0))
        self.t0, self.t1, self.w = t0, t1, w
        return w / math.sqrt(abs(t1 - t0))


def sample_dpmpp_sde_cfgpp(model, x, sigmas, eta=1.0, s_noise=1.0, noise_sampler=None, r=0.5, enable_multiscale=True,
                           multiscale_factor=0.5, multiscale_fullres_start=5, multiscale_fullres_end=8,
                           multiscale_intermittent_fullres=False, trace=None):
    """samplers.sample_dpmpp_sde_cfgpp (samplers.py:965-1254).  The manual post-CFG hook calls (:1137-1139, :1209-1211)
    return the guider's CFG output as "uncond_denoised" and reset old_uncond_denoised, so both cfg_denoised and
    cfg_denoised_2 equal the plain CFG outputs and the momentum terms never run."""
    n = len(sigmas) - 1
    if n < 1:
        return x
    ms = Multiscale(x.shape, n, enable_multiscale, multiscale_factor, multiscale_fullres_start,
                    multiscale_fullres_end, multiscale_intermittent_fullres)
    if noise_sampler is None:
        noise_sampler = BrownianIntervalNoise(x)
    sigma_fn = lambda t: (-t).exp()            # noqa: E731   (:1081-1085)
    t_fn = lambda sg: -sg.log()                # noqa: E731

    def denoise(xx, sigma, full):
        xp = xx if full else ms.down(xx)
        if trace is not None:
            trace.append(tuple(xp.shape[-2:]))
        d = model(xp, sigma)
        return d if full else ms.up(d)

    for i in range(n):
        full = ms.fullres(i)
        denoised = denoise(x, sigmas[i], full)
        if sigmas[i + 1] == 0:
            x = x + ((x - denoised) / sigmas[i]) * (sigmas[i + 1] - sigmas[i])                  # :1156-1159
            continue
        t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
        s = t + (t_next - t) * r
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(s), eta)
        s_ = t_fn(sd)
        x_2 = (sigma_fn(s_) / sigma_fn(t)) * x - (t - s_).expm1() * denoised + noise_sampler(sigma_fn(t), sigma_fn(s)) * s_noise * su
        denoised_2 = denoise(x_2, sigma_fn(s), full)                                             # same resolution flag (:1186)
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(t_next), eta)
        t_next_ = t_fn(sd)
        x = ((sigma_fn(t_next_) / sigma_fn(t)) * x
             - (t - t_next_).expm1() * ((1 - 1 / (2 * r)) * denoised + (1 / (2 * r)) * denoised_2)
             + noise_sampler(sigma_fn(t), sigma_fn(t_next)) * s_noise * su)
    return x


MULTISCALE_WHITELIST = ("dpmpp_sde_cfgpp", "sample_euler_ancestral", "sample_euler", "sample_dpmpp_2m_cfgpp")


def ksampler_sample(denoiser, seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, denoise=1.0,
                    enable_multiscale=True, multiscale_factor=0.5, multiscale_fullres_start=3,
                    multiscale_fullres_end=8, multiscale_intermittent_fullres=False, trace=None):
    """KSampler.sample -> common_ksampler -> sample1 -> CFGGuider.sample -> KSAMPLER.sample
    (sampling.py:773-1233, CFG.py:236-357).  denoiser(x[2B], sigma[2B], ctx[2B]) is apply_model."""
    denoise = denoise or 1.0                                                              # sampling.py:875
    latent_image = latent_image.float()
    generator = torch.manual_seed(seed)                                                   # ksampler_util.py:287-295
    noise = torch.randn(latent_image.size(), dtype=latent_image.dtype, layout=latent_image.layout,
                        generator=generator, device="cpu")
    sigmas = sigmas_for(scheduler, steps, denoise)
    if sampler_name == "dpmpp_2m_cfgpp":                                                  # sampling.py:517-532
        fn, disable_cfg1 = sample_dpmpp_2m_cfgpp, True
    elif sampler_name == "euler_ancestral_cfgpp":
        fn, disable_cfg1 = sample_euler_ancestral_cfgpp, True
    elif sampler_name == "euler_cfgpp":
        fn, disable_cfg1 = sample_euler_cfgpp, True
    elif sampler_name == "dpmpp_sde_cfgpp":
        fn, disable_cfg1 = sample_dpmpp_sde_cfgpp, True
    else:
        fn, disable_cfg1 = sample_euler, False
    extra = {}
    if sampler_name in MULTISCALE_WHITELIST:                                              # sampling.py:949-964
        extra = dict(enable_multiscale=enable_multiscale, multiscale_factor=multiscale_factor,
                     multiscale_fullres_start=multiscale_fullres_start, multiscale_fullres_end=multiscale_fullres_end,
                     multiscale_intermittent_fullres=multiscale_intermittent_fullres)
    if torch.count_nonzero(latent_image) > 0:                                             # CFG.py:266-269
        latent_image = latent_image * 0.18215
    max_sigma, s0 = float(SIGMAS[-1]), float(sigmas[0])                                   # sampling.py:410-422
    if math.isclose(max_sigma, s0, rel_tol=1e-05) or s0 > max_sigma:
        x = noise * torch.sqrt(1.0 + sigmas[0] ** 2.0)                                    # sampling.py:58-83
    else:
        x = noise * sigmas[0]
    x = x + latent_image

    def model(xx, sigma):
        return cfg_denoise(denoiser, xx, sigma, positive, negative, cfg, disable_cfg1)

    if fn is sample_euler_cfgpp:
        def model_uc(xx, sigma):                                                          # [uncond; cond] batched call
            bb = xx.shape[0]
            pos = positive.expand(bb, -1, -1) if positive.shape[0] == 1 else positive
            neg = negative.expand(bb, -1, -1) if negative.shape[0] == 1 else negative
            neg, pos = lcm_pad([neg, pos])
            out = denoiser(torch.cat([xx, xx]), sigma * xx.new_ones([2 * bb]), torch.cat([neg, pos]))
            return out.chunk(2)
        x = fn(model_uc, x, sigmas, cfg, trace=trace)
    else:
        x = fn(model, x, sigmas, trace=trace, **extra)
    return x / 0.18215                                                                    # CFG.py:294


# ---------------------------------------------------------------------------------------------------
# Latent upscale (src/Utilities/upscale.py:5-128), used by HiresFix (pipeline.py:346-350)
def _bislerp_data(length_old, length_new):
    """generate_bilinear_data (upscale.py:61-97)."""
    c1 = torch.arange(length_old, dtype=torch.float32).reshape(1, 1, 1, -1)
    c1 = F.interpolate(c1, size=(1, length_new), mode="bilinear")
    ratios = c1 - c1.floor()
    c1 = c1.to(torch.int64)
    c2 = torch.arange(length_old, dtype=torch.float32).reshape(1, 1, 1, -1) + 1
    c2[:, :, :, -1] -= 1
    c2 = F.interpolate(c2, size=(1, length_new), mode="bilinear").to(torch.int64)
    return ratios.reshape(-1), c1.reshape(-1), c2.reshape(-1)


def _slerp(b1, b2, r):
    """slerp (upscale.py:17-59) on [P, C] rows with ratio [P, 1]."""
    n1 = torch.norm(b1, dim=-1, keepdim=True)
    n2 = torch.norm(b2, dim=-1, keepdim=True)
    u1 = torch.where(n1 == 0.0, torch.zeros_like(b1), b1 / n1)
    u2 = torch.where(n2 == 0.0, torch.zeros_like(b2), b2 / n2)
    dot = (u1 * u2).sum(1)
    omega = torch.acos(dot)
    so = torch.sin(omega)
    res = (torch.sin((1.0 - r.squeeze(1)) * omega) / so).unsqueeze(1) * u1 + (torch.sin(r.squeeze(1) * omega) / so).unsqueeze(1) * u2
    res = res * (n1 * (1.0 - r) + n2 * r)
    res = torch.where((dot > 1 - 1e-5).unsqueeze(1), b1, res)
    res = torch.where((dot < 1e-5 - 1).unsqueeze(1), b1 * (1.0 - r) + b2 * r, res)
    return res


def bislerp(samples, width, height):
    """bislerp (upscale.py:5-128): slerp along W, then along H, of the channel vector at each pixel."""
    x = samples.float()
    n, c, h, w = x.shape
    r, c1, c2 = _bislerp_data(w, width)
    p1 = x[:, :, :, c1].movedim(1, -1).reshape(-1, c)
    p2 = x[:, :, :, c2].movedim(1, -1).reshape(-1, c)
    rr = r.reshape(1, 1, -1).expand(n, h, -1).reshape(-1, 1)
    x = _slerp(p1, p2, rr).reshape(n, h, width, c).movedim(-1, 1)
    r, c1, c2 = _bislerp_data(h, height)
    p1 = x[:, :, c1, :].movedim(1, -1).reshape(-1, c)
    p2 = x[:, :, c2, :].movedim(1, -1).reshape(-1, c)
    rr = r.reshape(1, -1, 1).expand(n, -1, width).reshape(-1, 1)
    x = _slerp(p1, p2, rr).reshape(n, height, width, c).movedim(-1, 1)
    return x.to(samples.dtype)


# ---------------------------------------------------------------------------------------------------
# VAE decode  (src/AutoEncoders/VariationalAE.py, src/AutoEncoders/ResBlock.py:341-406, src/Attention/Attention.py:127-178)
def vae_resnet_block(w, x):
    """ResnetBlock.forward (ResBlock.py:383-406): Normalize = GroupNorm(32, eps 1e-6) (Attention.py:11-31), swish."""
    h = F.conv2d(F.silu(group_norm(x, w, "norm1", 1e-6)), w("conv1.weight"), w("conv1.bias"), padding=1)
    h = F.conv2d(F.silu(group_norm(h, w, "norm2", 1e-6)), w("conv2.weight"), w("conv2.bias"), padding=1)
    if w.has("nin_shortcut.weight"):
        x = F.conv2d(x, w("nin_shortcut.weight"), w("nin_shortcut.bias"))
    return x + h


def vae_attn_block(w, x):
    """AttnBlock.forward (Attention.py:159-178) with pytorch_attention (AttentionMethods.py:175-197): one head of width C."""
    h = group_norm(x, w, "norm", 1e-6)
    q = F.conv2d(h, w("q.weight"), w("q.bias"))
    k = F.conv2d(h, w("k.weight"), w("k.bias"))
    v = F.conv2d(h, w("v.weight"), w("v.bias"))
    b, c, hh, ww = q.shape
    q, k, v = (t.view(b, 1, c, -1).transpose(2, 3).contiguous() for t in (q, k, v))
    out = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
    out = out.transpose(2, 3).reshape(b, c, hh, ww)
    return x + F.conv2d(out, w("proj_out.weight"), w("proj_out.bias"))


def vae_decode(sd, cfg, z):
    """VAE.decode (VariationalAE.py:690-722) -> AutoencodingEngine.decode (:130-145) -> Decoder.forward (:532-567);
    process_output clamp((x+1)/2, 0, 1) (:595-597); returns NHWC fp32."""
    w = W(sd)
    h = z.float()
    if cfg.use_post_quant:
        h = F.conv2d(h, w("post_quant_conv.weight"), w("post_quant_conv.bias"))
    d = w.sub("decoder.")
    h = F.conv2d(h, d("conv_in.weight"), d("conv_in.bias"), padding=1)
    h = vae_resnet_block(d.sub("mid.block_1."), h)
    h = vae_attn_block(d.sub("mid.attn_1."), h)
    h = vae_resnet_block(d.sub("mid.block_2."), h)
    nl = len(cfg.ch_mult)
    for lv in reversed(range(nl)):
        for i in range(cfg.num_res_blocks + 1):
            h = vae_resnet_block(d.sub(f"up.{lv}.block.{i}."), h)
        if lv != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")                        # Upsample VariationalAE.py:209-221
            h = F.conv2d(h, d(f"up.{lv}.upsample.conv.weight"), d(f"up.{lv}.upsample.conv.bias"), padding=1)
    h = F.silu(group_norm(h, d, "norm_out", 1e-6))
    h = F.conv2d(h, d("conv_out.weight"), d("conv_out.bias"), padding=1)
    return torch.clamp((h + 1.0) / 2.0, min=0.0, max=1.0).movedim(1, -1)


def vae_encode_moments(sd, cfg, pixels):
    """VAE.encode up to the regulariser (VariationalAE.py:725-760): pixels [B,H,W,3] in [0,1] -> process_input
    (x*2-1, :593) -> Encoder.forward (:378-413; Downsample = F.pad (0,1,0,1) + 3x3 stride-2 pad-0 conv, :224-254)
    -> quant_conv (:164-166) -> moments [B, 2*z, H/8, W/8] (mean | logvar)."""
    w = W(sd)
    e = w.sub("encoder.")
    x = pixels[..., :3].movedim(-1, 1).float() * 2.0 - 1.0
    h = F.conv2d(x, e("conv_in.weight"), e("conv_in.bias"), padding=1)
    nl = len(cfg.ch_mult)
    for lv in range(nl):
        for i in range(cfg.num_res_blocks):
            h = vae_resnet_block(e.sub(f"down.{lv}.block.{i}."), h)
        if lv != nl - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = F.conv2d(h, e(f"down.{lv}.downsample.conv.weight"), e(f"down.{lv}.downsample.conv.bias"), stride=2, padding=0)
    h = vae_resnet_block(e.sub("mid.block_1."), h)
    h = vae_attn_block(e.sub("mid.attn_1."), h)
    h = vae_resnet_block(e.sub("mid.block_2."), h)
    h = F.silu(group_norm(h, e, "norm_out", 1e-6))
    h = F.conv2d(h, e("conv_out.weight"), e("conv_out.bias"), padding=1)
    if cfg.use_post_quant:
        h = F.conv2d(h, w("quant_conv.weight"), w("quant_conv.bias"))
    return h


def vae_sample_moments(moments):
    """DiagonalGaussianRegularizer / DiagonalGaussianDistribution.sample (VariationalAE.py:34-51): logvar clamped
    to [-30, 20]; mean + exp(0.5 logvar) * torch.randn(mean.shape) from the global CPU RNG."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * torch.randn(mean.shape)


# ---------------------------------------------------------------------------------------------------
# CLIP text encoder  (src/clip/CLIPTextModel.py:51-107, src/clip/Clip.py:14-294, src/SD15/SDClip.py:36-97)
def clip_textual_embeddings(cfg, chunks, pad_token):
    """SDClipModel.set_up_textual_embeddings (src/SD15/SDClip.py:213-267): vector tokens of the model's width are appended to
    the token table and addressed as ids vocab_size, vocab_size + 1, ...; vectors of another width are dropped and the chunk
    is re-padded with the pad token.  Returns (integer chunks, [extra rows])."""
    nxt, rows, out = cfg.vocab_size, [], []
    for chunk in chunks:
        ids = []
        for t in chunk:
            if isinstance(t, int) or (hasattr(t, "ndim") and t.ndim == 0):
                ids.append(int(t))
            elif t.shape[0] == cfg.hidden_size:
                rows.append(torch.as_tensor(t).float()); ids.append(nxt); nxt += 1
        out.append(ids + [pad_token] * (len(chunk) - len(ids)))
    return out, rows


def clip_forward(sd, cfg, tokens, intermediate_output=None, final_layer_norm_intermediate=True, extra_rows=()):
    """CLIPTextModel_.forward: returns (x after final LN, intermediate (final-LN'd) or None, pooled).  extra_rows: rows
    appended to the token-embedding table for this call (textual inversion)."""
    w = W(sd)
    e, heads = cfg.hidden_size, cfg.num_heads
    table = w("embeddings.token_embedding.weight")
    if len(extra_rows):      # the reference's new nn.Embedding has the stored table's dtype (fp16 checkpoints round the vectors)
        stored = sd["embeddings.token_embedding.weight"].dtype
        table = torch.cat([table, torch.stack(list(extra_rows)).to(stored).to(table.dtype)], dim=0)
    x = F.embedding(tokens, table) + w("embeddings.position_embedding.weight")[: tokens.shape[1]]
    mask = torch.empty(x.shape[1], x.shape[1], dtype=x.dtype).fill_(float("-inf")).triu_(1)     # causal (CLIPTextModel.py:83-91)
    if intermediate_output is not None and intermediate_output < 0:
        intermediate_output = cfg.num_layers + intermediate_output                            # Clip.py:222-224
    inter = None
    for l in range(cfg.num_layers):
        lw = w.sub(f"encoder.layers.{l}.")
        n = F.layer_norm(x, (e,), lw("layer_norm1.weight"), lw("layer_norm1.bias"))
        q = F.linear(n, lw("self_attn.q_proj.weight"), lw("self_attn.q_proj.bias"))
        k = F.linear(n, lw("self_attn.k_proj.weight"), lw("self_attn.k_proj.bias"))
        v = F.linear(n, lw("self_attn.v_proj.weight"), lw("self_attn.v_proj.bias"))
        b = q.shape[0]
        qh, kh, vh = (t.view(b, -1, heads, e // heads).transpose(1, 2) for t in (q, k, v))
        a = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mask, dropout_p=0.0, is_causal=False)
        a = a.transpose(1, 2).reshape(b, -1, e)
        x = x + F.linear(a, lw("self_attn.out_proj.weight"), lw("self_attn.out_proj.bias"))
        n = F.layer_norm(x, (e,), lw("layer_norm2.weight"), lw("layer_norm2.bias"))
        hdn = F.linear(n, lw("mlp.fc1.weight"), lw("mlp.fc1.bias"))
        hdn = hdn * torch.sigmoid(1.702 * hdn)                                                # quick_gelu Clip.py:74-77
        x = x + F.linear(hdn, lw("mlp.fc2.weight"), lw("mlp.fc2.bias"))
        if l == intermediate_output:
            inter = x.clone()
    x = F.layer_norm(x, (e,), w("final_layer_norm.weight"), w("final_layer_norm.bias"))
    if inter is not None and final_layer_norm_intermediate:
        inter = F.layer_norm(inter, (e,), w("final_layer_norm.weight"), w("final_layer_norm.bias"))
    pos = (tokens == cfg.eos_token_id).int().argmax(dim=-1)                                   # CLIPTextModel.py:98-106
    pooled = x[torch.arange(x.shape[0]), pos]
    if w.has("text_projection.weight"):                                                       # CLIPTextModel.forward :146-150
        pooled = F.linear(pooled, w("text_projection.weight"))
    return x, inter, pooled


def clip_encode_token_weights(sd, cfg, token_weight_pairs, layer_idx=-2, special_tokens=(49406, 49407, 49407)):
    """ClipTokenWeightEncoder.encode_token_weights (SDClip.py:36-97) + SDClipModel.forward (:269-336)."""
    to_encode, has_weights, max_len = [], False, 0
    for x in token_weight_pairs:
        toks = [a[0] for a in x]
        max_len = max(max_len, len(toks))
        has_weights = has_weights or not all(a[1] == 1.0 for a in x)
        to_encode.append(toks)
    sections = len(to_encode)
    if has_weights or sections == 0:
        start, end, pad = special_tokens
        to_encode.append([start, end] + [pad] * (max_len - 2))
    to_encode, extra = clip_textual_embeddings(cfg, to_encode, special_tokens[2])
    last, inter, pooled = clip_forward(sd, cfg, torch.LongTensor(to_encode), intermediate_output=layer_idx, extra_rows=extra)
    out = (inter if layer_idx is not None else last).float()
    output = []
    for k in range(sections):
        z = out[k:k + 1].clone()
        if has_weights:
            z_empty = out[-1]
            for j in range(z.shape[1]):
                wgt = token_weight_pairs[k][j][1]
                if wgt != 1.0:
                    z[0][j] = (z[0][j] - z_empty[j]) * wgt + z_empty[j]
        output.append(z)
    cond = out[-1:] if not output else torch.cat(output, dim=-2)
    return cond, pooled[0:1]


# ---------------------------------------------------------------------------------------------------
# T5 encoder (src/clip/FluxClip.py): Flux's second text encoder
def t5_relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """T5Attention._relative_position_bucket, bidirectional (FluxClip.py:152-205)."""
    num_buckets //= 2
    rb = (relative_position > 0).to(torch.long) * num_buckets
    rp = torch.abs(relative_position)
    max_exact = num_buckets // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return rb + torch.where(is_small, rp, large)


def t5_rms_norm(x, weight, eps=1e-6):
    """T5LayerNorm.forward (FluxClip.py:565-582)."""
    return weight * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def t5_forward(sd, cfg, tokens):
    """T5.forward -> T5Stack.forward (FluxClip.py:441-519): relative-position bias computed by block 0 and reused,
    unscaled attention (k * sqrt(d) against SDPA's 1/sqrt(d), :265-268), gated tanh-GELU FF, final RMS norm."""
    w = W(sd)
    x = w("shared.weight")[tokens]
    b, l, e = x.shape
    h = cfg.num_heads
    d = e // h
    ctx = torch.arange(l, dtype=torch.long)[:, None]
    mem = torch.arange(l, dtype=torch.long)[None, :]
    bucket = t5_relative_position_bucket(mem - ctx, cfg.num_buckets, cfg.max_distance)
    bias = w("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight")[bucket].permute(2, 0, 1).unsqueeze(0)
    for i in range(cfg.num_layers):
        a, f = w.sub(f"encoder.block.{i}.layer.0."), w.sub(f"encoder.block.{i}.layer.1.")
        n = t5_rms_norm(x, a("layer_norm.weight"))
        q, k, v = (F.linear(n, a(f"SelfAttention.{t}.weight")).view(b, l, h, d).transpose(1, 2) for t in ("q", "k", "v"))
        o = F.scaled_dot_product_attention(q, k * (d ** 0.5), v, attn_mask=bias, dropout_p=0.0, is_causal=False)
        x = x + F.linear(o.transpose(1, 2).reshape(b, l, e), a("SelfAttention.o.weight"))
        n = t5_rms_norm(x, f("layer_norm.weight"))
        g = F.gelu(F.linear(n, f("DenseReluDense.wi_0.weight")), approximate="tanh") * F.linear(n, f("DenseReluDense.wi_1.weight"))
        x = x + F.linear(g, f("DenseReluDense.wo.weight"))
    return t5_rms_norm(x, w("encoder.final_layer_norm.weight"))


def t5_encode_token_weights(sd, cfg, token_weight_pairs):
    """ClipTokenWeightEncoder.encode_token_weights (SDClip.py:36-97) with T5XXLModel's special tokens {end 1, pad 0}."""
    to_encode, has_weights, max_len = [], False, 0
    for x in token_weight_pairs:
        toks = [a[0] for a in x]
        max_len = max(max_len, len(toks))
        has_weights = has_weights or not all(a[1] == 1.0 for a in x)
        to_encode.append(toks)
    sections = len(to_encode)
    if has_weights or sections == 0:
        to_encode.append([1] + [0] * (max_len - 1))
    out = t5_forward(sd, cfg, torch.tensor(to_encode, dtype=torch.int64))
    output = []
    for k in range(sections):
        z = out[k:k + 1].clone()
        if has_weights:
            z_empty = out[-1]
            for j in range(z.shape[1]):
                wgt = token_weight_pairs[k][j][1]
                if wgt != 1.0:
                    z[0][j] = (z[0][j] - z_empty[j]) * wgt + z_empty[j]
        output.append(z)
    return out[-1:] if not output else torch.cat(output, dim=-2)


# ---------------------------------------------------------------------------------------------------
# Flux DiT  (src/BlackForest/Flux.py)
def flux_rope(pos, dim, theta):
    """rope() (Flux.py:36-70): [..., n] -> [..., n, dim/2, 2, 2] rotation matrices, fp64 frequencies."""
    scale = torch.linspace(0, (dim - 2) / dim, steps=dim // 2, dtype=torch.float64)
    omega = 1.0 / (theta ** scale)
    out = torch.einsum("...n,d->...nd", pos.to(dtype=torch.float32), omega)
    out = torch.stack([torch.cos(out), -torch.sin(out), torch.sin(out), torch.cos(out)], dim=-1)
    return out.reshape(*out.shape[:-1], 2, 2).to(torch.float32)


def flux_apply_rope(xq, xk, freqs_cis):
    """apply_rope (Flux.py:73-82)."""
    xq_ = xq.float().reshape(*xq.shape[:-1], -1, 1, 2)
    xk_ = xk.float().reshape(*xk.shape[:-1], -1, 1, 2)
    xq_out = freqs_cis[..., 0] * xq_[..., 0] + freqs_cis[..., 1] * xq_[..., 1]
    xk_out = freqs_cis[..., 0] * xk_[..., 0] + freqs_cis[..., 1] * xk_[..., 1]
    return xq_out.reshape(*xq.shape), xk_out.reshape(*xk.shape)


def flux_temb(t, dim=256, max_period=10000, time_factor=1000.0):
    """timestep_embedding_flux (sample/sampling_util.py:78-104)."""
    t = time_factor * t
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _rms(x, scale):
    return F.rms_norm(x, scale.shape, weight=scale, eps=1e-6)                                  # rms_norm (Flux.py:502-524)


def _flux_attn(q, k, v, pe, mx=False):
    """attention() (Flux.py:18-33): rope then SDPA over [B,H,L,D], back to [B,L,H*D].  mx: the build's MX fp8 attention rule (mx_attention) instead of SDPA."""
    q, k = flux_apply_rope(q, k, pe)
    o = mx_attention(q, k, v) if mx else F.scaled_dot_product_attention(q, k, v)
    return o.transpose(1, 2).reshape(o.shape[0], o.shape[2], -1)


class FluxFBCache:
    """First-block cache state (WaveSpeed/fbcache_nodes.py:8-201 + first_block_cache.py:105-384) as the reference's Flux
    pipeline uses it (threshold 0.12, no validation function): reset when the input shape changes or the timestep does not
    decrease; hit when mean|r_prev - r| / mean|r_prev| < threshold for r = img_after_block0 - img_before."""

    def __init__(self, threshold):
        self.threshold, self.log = threshold, []
        self.reset()

    def reset(self):
        self.first = self.res_img = self.res_txt = None
        self.prev_t = self.prev_shape = None

    def begin(self, x, t):                                         # ensure_cache_state
        if self.prev_t is None or self.prev_shape != tuple(x.shape) or t >= self.prev_t:
            self.reset()
        self._cur = (tuple(x.shape), t)

    def end(self):                                                 # update_cache_state
        self.prev_shape, self.prev_t = self._cur

    def can_use(self, r):
        if self.first is None or self.first.shape != r.shape:
            return False
        return float((self.first - r).abs().mean() / self.first.abs().mean()) < self.threshold


def mx_fake_quant(t):
    """OCP microscaling fp8 along the last axis (a multiple of 32), returned DEQUANTISED in fp32: per block of 32 consecutive
    elements, scale = 2^ceil(log2(amax / 448)) (E8M0, exponent clamped to [-126, 126]), element = e4m3fn(x / scale) rounded
    to nearest even.  This is the build's MX fp8 mode (include/ldx.h: ldx_op_mx_quant / ldx_flux_set_fp8); the reference has
    no fp8 path (its Flux runs from Q8_0 weights dequantised to 16-bit, Quantizer.py:94-112), so this mode is NOT pinned by
    the reference: it is pinned by this rule, bit-for-bit at the op level (tests/test_mx_gpu.py)."""
    shp = t.shape
    xb = t.float().reshape(-1, shp[-1] // 32, 32)
    amax = xb.abs().amax(-1)
    r = amax * torch.tensor(np.float32(1.0) / np.float32(448.0))
    bits = r.view(torch.int32)
    e = (((bits >> 23) & 0xFF) + ((bits & 0x7FFFFF) != 0).to(torch.int32)).clamp(1, 253)
    inv = ((254 - e) << 23).view(torch.float32)
    scale = (e << 23).view(torch.float32)
    q = (xb * inv[..., None]).to(torch.float8_e4m3fn).float()
    return (q * scale[..., None]).reshape(shp)


MX_ATTN_PSH = 6          # P is rounded to e4m3 at the fixed scale 2^-6 (csrc/attn_mx.hip AM_PSH)


def mx_attn_key_of_k(k):
    """Key held by byte k (0..63) of a 64-key step of V8T: the contraction order of v_mfma_scale_f32_32x32x64_f8f6f4 as the P^T operand comes out of
    the S^T accumulators (csrc/attn_mx.hip header; measured by profiles/ubench/mx_layout32.hip)."""
    return 32 * (k >> 5) + 8 * ((k & 15) >> 2) + 4 * ((k >> 4) & 1) + (k & 3)


def mx_fake_quant_keys(v):
    """V of attention, [..., L, D]: MX fp8 along the KEY axis — one scale per (d, 32 consecutive keys), L padded with zeros to a multiple of 32 — dequantised."""
    L = v.shape[-2]
    Lp = (L + 31) // 32 * 32
    vt = torch.zeros(v.shape[:-2] + (v.shape[-1], Lp), dtype=torch.float32)
    vt[..., :L] = v.float().transpose(-1, -2)
    return mx_fake_quant(vt)[..., :L].transpose(-1, -2)


def mx_attention(q, k, v, scale=None):
    """The build's MX fp8 attention rule (include/ldx.h ldx_op_attention_fp8; no reference counterpart — Flux.py:18-33 is SDPA in 16 bit): q, k [B,H,L,D]
    quantised along d, v along the keys, T = q k^T * scale * log2(e), P = 2^(T - ceil(row max of T)) rounded to e4m3 at the fixed scale 2^-6,
    O = P8 v / sum of P8 (the rounded values: the kernel gets the row sums from an all-ones row of V^T).  The reference exponent is an INTEGER: the kernel's lazily updated one may sit up to two below it, i.e. its P is
    this P times 1, 2 or 4 — the same e4m3 mantissas (a floating format), only the subnormal boundary (2^-15 relative and below) moves."""
    d = q.shape[-1]
    scale = (1.0 / math.sqrt(d)) if scale is None else scale
    qf, kf, vf = mx_fake_quant(q.float()), mx_fake_quant(k.float()), mx_fake_quant_keys(v)
    t = (qf.double() @ kf.double().transpose(-1, -2)) * (np.float32(scale) * np.float32(1.44269504088896340736))
    p = torch.exp2(t - torch.ceil(t.amax(-1, keepdim=True)))
    p8 = (p * 2.0 ** MX_ATTN_PSH).float().to(torch.float8_e4m3fn).double() / 2.0 ** MX_ATTN_PSH
    return ((p8 @ vf.double()) / p8.sum(-1, keepdim=True)).float()          # the denominator sums the ROUNDED P (a ones row of V^T on the matrix pipe)


# the linears of the build's MX fp8 mode: the block linears and (round 6) the adaLN modulation projections, which the engine runs as one batched skinny GEMM on MX weights
_MX_LINEARS = ("_attn.qkv", "_attn.proj", "_mlp.0", "_mlp.2", ".linear1", ".linear2", "_mod.lin", ".modulation.lin", "final_layer.adaLN_modulation.1")


def flux_forward(sd, cfg, x, timestep, context, y, guidance, fb=None, mx=False, mx_attn=None):
    """Flux3.forward + forward_orig (Flux.py:658-778).  x [B,C,h,w] (h, w even), returns the raw model output.
    fb: optional FluxFBCache (approximate mode).  mx: the build's MX fp8 mode — input and weight of every double / single
    block linear pass through mx_fake_quant (fp32 accumulation, everything else unchanged); mx_attn (default False = ldx_flux_set_fp8 mode 1,
    linears only): True = mode 3, the joint attention follows mx_attention as well (head dim 128)."""
    w = W(sd)
    if mx_attn is None:
        mx_attn = False

    def lin(name, t):
        wt = w(name + ".weight")
        if mx and name.endswith(_MX_LINEARS):
            t, wt = mx_fake_quant(t), mx_fake_quant(wt)
        return F.linear(t, wt, w(name + ".bias") if w.has(name + ".bias") else None)

    mlp_emb = lambda name, t: lin(name + ".out_layer", F.silu(lin(name + ".in_layer", t)))
    bs, c, h, wd = x.shape
    hl, wl = h // 2, wd // 2
    img = x.reshape(bs, c, hl, 2, wl, 2).permute(0, 2, 4, 1, 3, 5).reshape(bs, hl * wl, c * 4)    # b (h w) (c ph pw)
    img_ids = torch.zeros((hl, wl, 3))
    img_ids[..., 1] += torch.linspace(0, hl - 1, steps=hl)[:, None]
    img_ids[..., 2] += torch.linspace(0, wl - 1, steps=wl)[None, :]
    img_ids = img_ids.reshape(1, hl * wl, 3).expand(bs, -1, -1)
    txt_ids = torch.zeros((bs, context.shape[1], 3))
    ids = torch.cat((txt_ids, img_ids), dim=1)
    pe = torch.cat([flux_rope(ids[..., i], cfg.axes_dim[i], cfg.theta) for i in range(3)], dim=-3).unsqueeze(1)
    img = lin("img_in", img)
    vec = mlp_emb("time_in", flux_temb(timestep))
    if cfg.guidance_embed:
        vec = vec + mlp_emb("guidance_in", flux_temb(guidance))
    vec = vec + mlp_emb("vector_in", y)
    txt = lin("txt_in", context)
    C, H = cfg.hidden_size, cfg.num_heads
    ln = lambda t: F.layer_norm(t, (C,), eps=1e-6)
    heads = lambda t: t.view(t.shape[0], t.shape[1], 3, H, -1).permute(2, 0, 3, 1, 4)
    lt = txt.shape[1]
    if fb is not None:
        fb.begin(x, float(timestep[0]))
    img_in, skip_rest = img, False
    for i in range(cfg.depth):                                                                 # DoubleStreamBlock.forward :298-348
        if skip_rest:
            break
        p = f"double_blocks.{i}."
        im = lin(p + "img_mod.lin", F.silu(vec))[:, None, :].chunk(6, dim=-1)
        tm = lin(p + "txt_mod.lin", F.silu(vec))[:, None, :].chunk(6, dim=-1)
        iq, ik, iv = heads(lin(p + "img_attn.qkv", (1 + im[1]) * ln(img) + im[0]))
        iq, ik = _rms(iq, w(p + "img_attn.norm.query_norm.scale")), _rms(ik, w(p + "img_attn.norm.key_norm.scale"))
        tq, tk, tv = heads(lin(p + "txt_attn.qkv", (1 + tm[1]) * ln(txt) + tm[0]))
        tq, tk = _rms(tq, w(p + "txt_attn.norm.query_norm.scale")), _rms(tk, w(p + "txt_attn.norm.key_norm.scale"))
        attn = _flux_attn(torch.cat((tq, iq), 2), torch.cat((tk, ik), 2), torch.cat((tv, iv), 2), pe, mx_attn)
        ta, ia = attn[:, :lt], attn[:, lt:]
        img = img + im[2] * lin(p + "img_attn.proj", ia)
        img = img + im[5] * lin(p + "img_mlp.2", F.gelu(lin(p + "img_mlp.0", (1 + im[4]) * ln(img) + im[3]), approximate="tanh"))
        txt = txt + tm[2] * lin(p + "txt_attn.proj", ta)
        txt = txt + tm[5] * lin(p + "txt_mlp.2", F.gelu(lin(p + "txt_mlp.0", (1 + tm[4]) * ln(txt) + tm[3]), approximate="tanh"))
        if fb is not None and i == 0:                                                          # CachedTransformerBlocks.forward :253-330
            r = img - img_in
            if fb.can_use(r):
                fb.log.append(1)
                img, txt, skip_rest = img + fb.res_img, txt + fb.res_txt, True
            else:
                fb.log.append(0)
                fb.first, img_b0, txt_b0 = r, img, txt
    xj = torch.cat((txt, img), 1)
    for i in range(cfg.depth_single_blocks):                                                   # SingleStreamBlock.forward :389-418
        if skip_rest:
            break
        p = f"single_blocks.{i}."
        shift, scale, gate = lin(p + "modulation.lin", F.silu(vec))[:, None, :].chunk(3, dim=-1)
        qkv, mlp = torch.split(lin(p + "linear1", (1 + scale) * ln(xj) + shift), [3 * C, cfg.mlp_hidden], dim=-1)
        q, k, v = heads(qkv)
        q, k = _rms(q, w(p + "norm.query_norm.scale")), _rms(k, w(p + "norm.key_norm.scale"))
        attn = _flux_attn(q, k, v, pe, mx_attn)
        xj = xj + gate * lin(p + "linear2", torch.cat((attn, F.gelu(mlp, approximate="tanh")), 2))
    img = xj[:, lt:]
    if fb is not None:
        if not skip_rest:
            fb.res_img, fb.res_txt = img - img_b0, xj[:, :lt] - txt_b0
        fb.end()
    shift, scale = lin("final_layer.adaLN_modulation.1", F.silu(vec)).chunk(2, dim=1)           # LastLayer.forward :455-471
    img = lin("final_layer.linear", (1 + scale[:, None, :]) * ln(img) + shift[:, None, :])
    return img.reshape(bs, hl, wl, c, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(bs, c, h, wd)


def flux_apply_model(sd, cfg, x, sigma, context, y, guidance, fb=None):
    """BaseModel.apply_model with CONST (sampling.py:100-155): input unscaled, t = sigma, denoised = x - out*sigma."""
    out = flux_forward(sd, cfg, x, sigma, context, y, guidance, fb=fb)
    return x - out * sigma.view(-1, 1, 1, 1)


# ---------------------------------------------------------------------------------------------------
# Flux sampling path: ModelSamplingFlux + CONST (sampling.py:100-218), Flux1 latent format (Latent.py:114-161)
def flux_time_shift(mu, sigma, t):
    """sampling.py:158-169."""
    return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)


def flux_model_sigmas(shift=1.15, timesteps=10000):
    """ModelSamplingFlux.set_parameters (sampling.py:183-195)."""
    return flux_time_shift(shift, 1.0, torch.arange(1, timesteps + 1, 1) / timesteps)


FLUX_SCALE, FLUX_SHIFT = 0.3611, 0.1159


def flux_ksampler_sample(denoiser, seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, guidance=3.0,
                         denoise=1.0, shift=1.15, enable_multiscale=True, multiscale_factor=0.5, multiscale_fullres_start=3,
                         multiscale_fullres_end=8, multiscale_intermittent_fullres=False, trace=None):
    """KSampler.sample(flux=True) (sampling.py:773-1233, CFG.py:236-357) for one positive / one negative prompt.
    positive / negative = (ctx [1,Lt,C], pooled [1,V]); denoiser(x, sigma[B], ctx, y, guidance[B]) is apply_model.
    CONST: x0 = sigma0*noise + (1-sigma0)*latent, result / (1 - sigma_last); latent format (z - shift)*scale in (non-empty
    latents only, CFG.py:266-269), z/scale + shift out."""
    denoise = denoise or 1.0
    latent_image = latent_image.float()
    generator = torch.manual_seed(seed)
    noise = torch.randn(latent_image.size(), dtype=latent_image.dtype, layout=latent_image.layout, generator=generator, device="cpu")
    table = flux_model_sigmas(shift)
    sigmas = sigmas_for(scheduler, steps, denoise, table)
    if torch.count_nonzero(latent_image) > 0:
        latent_image = (latent_image - FLUX_SHIFT) * FLUX_SCALE
    x = sigmas[0] * noise + (1.0 - sigmas[0]) * latent_image
    b = x.shape[0]
    (pc, py), (nc, ny) = positive, negative

    def model_uc(xx, sigma):
        bb = xx.shape[0]
        sig = sigma * xx.new_ones([2 * bb])
        out = denoiser(torch.cat([xx, xx]), sig, torch.cat([nc.expand(bb, -1, -1), pc.expand(bb, -1, -1)]),
                       torch.cat([ny.expand(bb, -1), py.expand(bb, -1)]), torch.full([2 * bb], float(guidance)))
        return out.chunk(2)

    def model(xx, sigma):
        if math.isclose(cfg, 1.0):
            bb = xx.shape[0]
            return denoiser(xx, sigma * xx.new_ones([bb]), pc.expand(bb, -1, -1), py.expand(bb, -1), torch.full([bb], float(guidance)))
        u, c = model_uc(xx, sigma)
        return torch.lerp(u, c, cfg)

    if sampler_name == "euler_cfgpp":
        x = sample_euler_cfgpp(model_uc, x, sigmas, cfg, trace=trace)
    elif sampler_name in ("dpmpp_2m_cfgpp", "euler_ancestral_cfgpp", "dpmpp_sde_cfgpp"):
        raise NotImplementedError(sampler_name)
    else:                                                                                 # unknown names fall back to sample_euler
        extra = {}
        if sampler_name in MULTISCALE_WHITELIST:                                          # sampling.py:949-964
            extra = dict(enable_multiscale=enable_multiscale, multiscale_factor=multiscale_factor,
                         multiscale_fullres_start=multiscale_fullres_start, multiscale_fullres_end=multiscale_fullres_end,
                         multiscale_intermittent_fullres=multiscale_intermittent_fullres)
        x = sample_euler(model, x, sigmas, trace=trace, **extra)
    x = x / (1.0 - sigmas[-1])                                                            # inverse_noise_scaling
    return x / FLUX_SCALE + FLUX_SHIFT


# ---------------------------------------------------------------------------------------------------
# ESRGAN RRDBNet + tiled_scale (src/UltimateSDUpscale/RDRB.py:10-471, USDU_util.py:36-138, Utilities/util.py:406-640)
def rrdbnet_forward(sd, cfg, x):
    """RRDBNet.forward on NCHW fp32; keys = the module's own names (model.0, model.1.sub.i.RDBk.convj.0, ...)."""
    w = W(sd)
    conv = lambda name, t: F.conv2d(t, w(name + ".weight"), w(name + ".bias"), padding=1)      # noqa: E731
    lrelu = lambda t: F.leaky_relu(t, 0.2)                                                        # noqa: E731
    fea = conv("model.0", x.float())
    h = fea
    for i in range(cfg.num_blocks):
        rin = h
        for k in (1, 2, 3):                                                                       # ResidualDenseBlock_5C.forward :199-205
            p = f"model.1.sub.{i}.RDB{k}.conv"
            x1 = lrelu(conv(p + "1.0", h))
            x2 = lrelu(conv(p + "2.0", torch.cat((h, x1), 1)))
            x3 = lrelu(conv(p + "3.0", torch.cat((h, x1, x2), 1)))
            x4 = lrelu(conv(p + "4.0", torch.cat((h, x1, x2, x3), 1)))
            x5 = conv(p + "5.0", torch.cat((h, x1, x2, x3, x4), 1))
            h = x5 * 0.2 + h
        h = h * 0.2 + rin                                                                         # RRDB.forward :72-76
    h = fea + conv(f"model.1.sub.{cfg.num_blocks}", h)                                            # ShortcutBlock
    nu = int(math.log2(cfg.scale))
    for u in range(nu):
        h = lrelu(conv(f"model.{3 * (u + 1)}", F.interpolate(h, scale_factor=2, mode="nearest")))
    h = lrelu(conv(f"model.{3 * nu + 2}", h))
    return conv(f"model.{3 * nu + 4}", h)


def tiled_scale(samples, function, tile_x=64, tile_y=64, overlap=8, upscale_amount=4, out_channels=3):
    """tiled_scale -> tiled_scale_multidim (util.py:406-640) for 2-D tiles, NCHW."""
    tile = (tile_y, tile_x)
    output = torch.empty([samples.shape[0], out_channels, round(samples.shape[2] * upscale_amount), round(samples.shape[3] * upscale_amount)])
    for b in range(samples.shape[0]):
        s = samples[b:b + 1]
        if all(s.shape[d + 2] <= tile[d] for d in range(2)):
            output[b:b + 1] = function(s)
            continue
        out = torch.zeros([1, out_channels] + list(output.shape[2:]))
        out_div = torch.zeros_like(out)
        positions = [range(0, s.shape[d + 2] - overlap, tile[d] - overlap) if s.shape[d + 2] > tile[d] else [0] for d in range(2)]
        for y in positions[0]:
            for x in positions[1]:
                s_in, up = s, []
                for d, it in ((0, y), (1, x)):
                    pos = max(0, min(s.shape[d + 2] - overlap, it))
                    ln = min(tile[d], s.shape[d + 2] - pos)
                    s_in = s_in.narrow(d + 2, pos, ln)
                    up.append(round(pos * upscale_amount))
                ps = function(s_in)
                mask = torch.ones_like(ps)
                feather = round(overlap * upscale_amount)
                for d in (2, 3):
                    if feather >= mask.shape[d]:
                        continue
                    for t in range(feather):
                        a = (t + 1) / feather
                        mask.narrow(d, t, 1).mul_(a)
                        mask.narrow(d, mask.shape[d] - 1 - t, 1).mul_(a)
                o = out.narrow(2, up[0], mask.shape[2]).narrow(3, up[1], mask.shape[3])
                od = out_div.narrow(2, up[0], mask.shape[2]).narrow(3, up[1], mask.shape[3])
                o.add_(ps * mask)
                od.add_(mask)
        output[b:b + 1] = out / out_div
    return output

