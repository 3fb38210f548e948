"""Step-invariant work cached by the engine (round 5), both bit-identical to recomputing:

* the per-timestep table of the ResBlocks' emb_layers outputs (time_embed MLP -> SiLU -> 22 emb_layers: a pure function of the INTEGER timestep,
  unet.py:333-342 / ResBlock.py:283-295 / sampling.py:309-320) — built at ldx_finalize by the same skinny kernels, gathered by the prep kernel;
  LDX_EMB_TABLE=0 (read once per process) restores the three per-step launches;
* the context cache (ldx_unet_context_cache): the 16-bit copy of c_crossattn and the batched to_k | to_v projections of all cross-attentions
  (Attention.py:100-124) once per (input shape, ctx buffer) instead of once per step, with explicit invalidation.
"""
import hashlib
import os
import subprocess
import sys
import textwrap

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng(ldx, ldx_lib):
    cfg = ldx.UNetConfig.tiny(64, 128)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    return cfg, ldx.UNetEngine(cfg, sd, device=0, dtype="bf16")


_CODE = textwrap.dedent('''
    import sys, hashlib, torch
    sys.path.insert(0, %r)
    import ldx_amd as ldx
    cfg = ldx.UNetConfig.tiny(64, 128)
    sd = ldx.weights.synth_state_dict(ldx.weights.unet_state_dict_spec(cfg), seed=1234)
    g = torch.Generator().manual_seed(11)
    x = torch.randn([3, 4, 16, 16], generator=g).cuda(); ctx = torch.randn([3, 77, 128], generator=g).cuda()
    h = hashlib.sha256()
    for dt in ("bf16", "f16"):
        eng = ldx.UNetEngine(cfg, sd, device=0, dtype=dt)
        for sig in ([14.6, 3.0, 0.03], [0.5, 0.5, 7.7]):
            h.update(eng.denoise(x, torch.tensor(sig).cuda(), ctx).cpu().numpy().tobytes())
        h.update(eng.forward(x, torch.tensor([0.0, 500.0, 999.0]).cuda(), ctx).cpu().numpy().tobytes())
        print("launches", dt, eng.plan_info()["launches"])
    print("SHA", h.hexdigest())
''') % ROOT


def test_emb_table_is_bit_identical_to_the_per_step_mlp(ldx_lib):
    res = {}
    for mode in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", _CODE], env=dict(os.environ, LDX_EMB_TABLE=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = ([l for l in r.stdout.splitlines() if l.startswith("SHA")][0], [int(l.split()[-1]) for l in r.stdout.splitlines() if l.startswith("launches")])
    assert res["1"][0] == res["0"][0], res                       # same bits from every timestep incl. the table's first and last row
    assert all(a == b - 3 for a, b in zip(res["1"][1], res["0"][1])), res      # time_embed.0, time_embed.2, emb_layers: three launches gone


def _sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


@pytest.mark.parametrize("graph", [False, True])
def test_context_cache_equals_recompute_and_follows_rewrites(eng, ldx, graph):
    cfg, e = eng
    gen = torch.Generator().manual_seed(3)
    x = torch.randn([2, 4, 16, 16], generator=gen).cuda()
    ctxA = torch.randn([2, 154, cfg.context_dim], generator=gen).cuda()
    ctxB = torch.randn([2, 154, cfg.context_dim], generator=gen).cuda()
    sig = [torch.tensor([s, s]).cuda() for s in (9.0, 2.0, 0.4)]
    e.set_graph_mode(False)
    ref = {n: [e.denoise(x, s, c).clone() for s in sig] for n, c in (("A", ctxA), ("B", ctxB))}      # default: recomputed every call
    buf = ctxA.clone()
    out = torch.empty_like(x)
    e.set_graph_mode(graph)
    try:
        n_before = e.plan_info()["launches"]
        for rep in range(3):                                       # steps of a "run" on one buffer: projections computed on the first call only
            for i, s in enumerate(sig):
                assert torch.equal(e.denoise(x, s, buf, out=out, ctx_cached=True), ref["A"][i]), (rep, i)
        assert e.plan_info()["launches"] == n_before - 2           # ctx.cvt + the batched k|v GEMM are not part of a steady-state step
        buf.copy_(ctxB)                                            # rewrite in place ...
        e.invalidate_context()                                     # ... and say so
        for i, s in enumerate(sig):
            assert torch.equal(e.denoise(x, s, buf, out=out, ctx_cached=True), ref["B"][i]), i
        # the default entry point never trusts a pointer: same buffer, new contents, no invalidate call
        buf.copy_(ctxA)
        for i, s in enumerate(sig):
            assert torch.equal(e.denoise(x, s, buf, out=out), ref["A"][i]), i
        assert e.plan_info()["launches"] == n_before
        # another input shape and back (plan cache): each plan keeps its own projections
        buf.copy_(ctxB); e.invalidate_context()
        x2 = torch.randn([2, 4, 8, 8], generator=gen).cuda()
        o_small = [e.denoise(x2, sig[0], buf, ctx_cached=True).clone() for _ in range(2)]
        assert torch.equal(o_small[0], o_small[1])
        assert torch.equal(e.denoise(x, sig[1], buf, out=out, ctx_cached=True), ref["B"][1])
        assert torch.equal(e.denoise(x2, sig[0], buf, ctx_cached=True), o_small[0])
    finally:
        e.set_graph_mode(False)
        e.set_context_cache(False)


def test_cfg_denoisers_sharing_a_buffer_do_not_see_each_others_projections(eng, ldx):
    """Two sampling runs of the same shape on one engine share the context buffer (sampling.CFGDenoiser's pool): the second run's prompts must
    replace the first run's cached projections, and going back must restore them (ownership hand-over goes through _take_context)."""
    cfg, e = eng
    gen = torch.Generator().manual_seed(8)
    mk = lambda: (torch.randn([1, 77, cfg.context_dim], generator=gen), torch.randn([1, 77, cfg.context_dim], generator=gen))
    (p1, n1), (p2, n2) = mk(), mk()
    x = torch.randn([1, 4, 16, 16], generator=gen).cuda()
    want = {}
    for name, (p, n) in (("1", (p1, n1)), ("2", (p2, n2))):
        ctx = torch.cat([n, p]).cuda()
        want[name] = e.denoise(torch.cat([x, x]), torch.tensor([3.0, 3.0]).cuda(), ctx).clone()
    d1 = ldx.sampling.CFGDenoiser(e, p1, n1, 7.0, 1, 16, 16)
    a = torch.cat(d1(x, torch.tensor(3.0))).clone()
    d2 = ldx.sampling.CFGDenoiser(e, p2, n2, 7.0, 1, 16, 16)
    assert d2.ctx.data_ptr() == d1.ctx.data_ptr()                # shared buffer: the interesting case
    b = torch.cat(d2(x, torch.tensor(3.0))).clone()
    c = torch.cat(d1(x, torch.tensor(3.0))).clone()              # d1 takes the buffer back
    assert torch.equal(a, want["1"]) and torch.equal(b, want["2"]) and torch.equal(c, want["1"])
    import gc
    del d1
    gc.collect()                                                 # the pool holds its owner weakly: nothing keeps a finished run's denoiser alive
    assert torch.equal(torch.cat(d2(x, torch.tensor(3.0))), want["2"])
    # clear_pool: the parked buffers go, denoisers created before keep theirs, a new one allocates afresh and is still right
    assert ldx.sampling.CFGDenoiser.clear_pool(e) > 0 and ldx.sampling.CFGDenoiser.clear_pool(e) == 0
    d3 = ldx.sampling.CFGDenoiser(e, p1, n1, 7.0, 1, 16, 16)
    assert d3.ctx.data_ptr() != d2.ctx.data_ptr()
    assert torch.equal(torch.cat(d3(x, torch.tensor(3.0))), want["1"])
    assert torch.equal(torch.cat(d2(x, torch.tensor(3.0))), want["2"])
    e.set_context_cache(False)
