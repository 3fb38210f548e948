"""Golden capture for the T5 text encoder by importing the reference's T5 / T5XXLModel classes with a
tiny config.  Build container only; writes tests/golden/t5.npz.  See oracle/ref_capture.py."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import ref_capture  # noqa: E402


def main():
    sys.path.insert(0, REPO)
    import ldx_amd as ldx
    torch.set_num_threads(8)
    ref_capture.enter_reference()
    OUT = ref_capture.OUT
    from src.clip import FluxClip
    from src.SD15 import SDClip
    from src.cond import cast
    cfg = ldx.T5Config.tiny()
    sd = ldx.weights.synth_state_dict(ldx.weights.t5_state_dict_spec(cfg), seed=555)
    g = {}
    gen = torch.Generator().manual_seed(3)
    # raw model: T5(config, dtype, device, operations)
    model = FluxClip.T5(cfg.reference_dict(), torch.float32, "cpu", cast.manual_cast)
    model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    model.eval()
    for name, (b, l) in (("a", (2, 64)), ("b", (1, 256)), ("c", (3, 40)), ("d", (1, 300))):
        ids = torch.randint(2, cfg.vocab_size, (b, l), generator=gen)
        ids[:, l - l // 4:] = 0
        ids[:, l - l // 4 - 1] = 1
        with torch.no_grad():
            out, inter = model(ids, dtype=torch.float32)
        g[f"ids_{name}"] = ids.numpy(); g[f"out_{name}"] = out.float().numpy()
    # relative-position buckets and bias, as compute_bias builds them
    att = model.encoder.block[0].layer[0].SelfAttention
    for l in (8, 40, 256, 300):
        ctx = torch.arange(l)[:, None]; mem = torch.arange(l)[None, :]
        g[f"bucket_{l}"] = att._relative_position_bucket(mem - ctx, True, 32, 128).numpy()
    g["bias_40"] = att.compute_bias(40, 40, "cpu", torch.float32).detach().numpy()
    # the SDClipModel wrapper with token weights (T5XXLModel with the tiny config injected)
    path = "/tmp/ldx_ref_scratch/t5_tiny.json"
    json.dump(cfg.reference_dict(), open(path, "w"))
    wrap = SDClip.SDClipModel(device="cpu", layer="last", layer_idx=None, textmodel_json_config=path, dtype=torch.float32,
                              special_tokens={"end": 1, "pad": 0}, model_class=FluxClip.T5, model_options={})
    wrap.transformer.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    toks = [int(t) for t in torch.randint(2, cfg.vocab_size, (30,), generator=gen)] + [1] + [0] * 33
    wts = [1.0] * 64
    wts[3], wts[4], wts[10] = 1.3, 1.3, 0.7
    pairs = [list(zip(toks, wts))]
    with torch.no_grad():
        cond, pooled = wrap.encode_token_weights(pairs)[:2]
    assert pooled is None
    g["tw_ids"] = np.array(toks); g["tw_wts"] = np.array(wts, dtype=np.float32); g["tw_cond"] = cond.float().numpy()
    np.savez_compressed(os.path.join(OUT, "t5.npz"), **g)
    print("t5.npz", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
