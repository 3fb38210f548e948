"""Golden capture for the Flux DiT (G11): tiny Flux3 through the reference's own module (cast.manual_cast ops).
Build container only; writes tests/golden/flux.npz."""
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
    from src.BlackForest import Flux
    from src.cond import cast
    cfg = ldx.FluxConfig.tiny()
    model = Flux.Flux3(dtype=torch.float32, device=torch.device("cpu"), operations=cast.manual_cast, **cfg.reference_kwargs())
    sd = ldx.weights.synth_state_dict(ldx.weights.flux_state_dict_spec(cfg), seed=31, dtype=torch.float32)
    res = model.load_state_dict(sd, strict=True)
    gen = torch.Generator().manual_seed(5)
    g = {}
    for name, (b, h, w, lt) in {"a": (1, 6, 10, 7), "b": (2, 8, 8, 16)}.items():
        x = torch.randn([b, 16, h, w], generator=gen)
        ctx = torch.randn([b, lt, cfg.context_in_dim], generator=gen)
        y = torch.randn([b, cfg.vec_in_dim], generator=gen)
        t = torch.rand([b], generator=gen) * 0.9 + 0.05
        gd = torch.full([b], 3.0)
        with torch.no_grad():
            out = model(x, t, ctx, y, gd)
        g.update({f"{name}_x": x.numpy(), f"{name}_ctx": ctx.numpy(), f"{name}_y": y.numpy(), f"{name}_t": t.numpy(),
                  f"{name}_g": gd.numpy(), f"{name}_out": out.float().numpy()})
    pe = model.pe_embedder(torch.tensor([[[0., 0., 0.], [0., 1., 2.], [0., 2., 1.]]]))
    g["pe_probe"] = pe.numpy()
    np.savez_compressed(os.path.join(ref_capture.OUT, "flux.npz"), **g)
    print("flux.npz", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
