"""The MX fp8 attention RULE on the CPU (oracle.mx_attention, the checker of tests/test_attn_mx_gpu.py): self-consistency of the restatement — the key
permutation is a bijection of every 64-key step, quantising along the keys equals quantising the transposed tensor, and the rule stays within the
stated distance of exact attention on the dequantised operands (what rounding P to e4m3 costs)."""
import math

import torch

from oracle import sd15_oracle as O


def test_key_permutation_is_a_bijection_of_a_64_key_step():
    keys = [O.mx_attn_key_of_k(k) for k in range(64)]
    assert sorted(keys) == list(range(64))
    # scale blocks: hardware k 32 t .. 32 t + 31 <-> the 32 consecutive keys of tile t
    assert sorted(keys[:32]) == list(range(32)) and sorted(keys[32:]) == list(range(32, 64))


def test_quantising_along_the_keys_is_mx_on_the_transpose():
    g = torch.Generator().manual_seed(2)
    v = torch.randn(2, 3, 70, 128, generator=g).bfloat16().float()
    got = O.mx_fake_quant_keys(v)
    pad = torch.zeros(2, 3, 128, 96)
    pad[..., :70] = v.transpose(-1, -2)
    assert torch.equal(got, O.mx_fake_quant(pad)[..., :70].transpose(-1, -2))
    assert float((got - v).abs().max() / v.abs().max()) < 0.07          # e4m3: 3 mantissa bits


def test_rule_is_close_to_exact_attention_of_the_dequantised_operands():
    g = torch.Generator().manual_seed(4)
    q, k, v = (torch.randn(1, 2, 96, 128, generator=g).bfloat16().float() for _ in range(3))
    rule = O.mx_attention(q, k, v)
    qf, kf, vf = O.mx_fake_quant(q), O.mx_fake_quant(k), O.mx_fake_quant_keys(v)
    exact = torch.softmax(qf.double() @ kf.double().transpose(-1, -2) / math.sqrt(128), -1) @ vf.double()
    r = float((rule.double() - exact).norm() / exact.norm())
    full = torch.softmax(q.double() @ k.double().transpose(-1, -2) / math.sqrt(128), -1) @ v.double()
    rq = float((rule.double() - full).norm() / full.norm())
    print(f"rule vs exact on dequantised operands {r:.3e}; vs un-quantised attention {rq:.3e}")
    assert r < 3e-2 and rq < 8e-2          # 96 keys: little averaging of the 3-bit P mantissas (measured 2.2e-2 / 5.2e-2); long sequences are far closer (test_attn_mx_gpu.py)
