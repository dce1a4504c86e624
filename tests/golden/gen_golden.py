#!/usr/bin/env python3
"""Generate tests/golden/*.npz from torch-CPU scaled_dot_product_attention.

Run in the build container only (needs torch; reads nothing from /root/reference
at run time -- the seeds, shapes, scales and tolerances below restate the
reference's own parity tests, cited per case).  torch SDPA is the reference's
declared ground truth: examples/pytorch-custom-op-ffi/tests/conftest.py:165-182,
tests/test_scale_factor_fix.py:55-66, Tests/test_mfa_systematic.py:56-61.

Fixtures are data only: inputs (fp32, or fp16 / bf16-bit arrays) and expected
outputs computed in fp64 by torch on exactly those (already rounded) inputs.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from oracle import oracle  # noqa: E402  (lcg generator + bf16 helpers only)


def sdpa64(q, k, v, **kw):
    """fp64 torch SDPA on the given (already dtype-rounded) tensors -> fp32."""
    with torch.inference_mode():
        m = kw.pop("attn_mask", None)
        if m is not None and m.dtype != torch.bool:
            m = m.double()
        o = F.scaled_dot_product_attention(q.double(), k.double(), v.double(), attn_mask=m, **kw)
    return o.float().numpy()


def sdpa32(q, k, v, **kw):
    with torch.inference_mode():
        return F.scaled_dot_product_attention(q, k, v, **kw).numpy()


def store_bits(t: torch.Tensor) -> np.ndarray:
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def main() -> None:
    torch.set_num_threads(4)

    # -- 1. scale-factor sweep, test_scale_factor_fix.py:33-66 (fp32, tol 1e-5) -----------
    out = {}
    for s, d in [(4, 4), (8, 8), (16, 16), (32, 32)]:
        torch.manual_seed(42)
        q, k, v = (torch.randn(s, d) for _ in range(3))
        out[f"q_{s}"], out[f"k_{s}"], out[f"v_{s}"] = q.numpy(), k.numpy(), v.numpy()
        for sc in [0.1, 0.25, 0.35355, 0.5, 1.0]:
            out[f"o_{s}_{sc}"] = sdpa32(q, k, v, scale=sc)
            out[f"o64_{s}_{sc}"] = sdpa64(q, k, v, scale=sc)
        out[f"o_{s}_default"] = sdpa32(q, k, v)  # default scale == 1/sqrt(D), :68-98
    np.savez_compressed(HERE / "scale_sweep_fp32.npz", **out)

    # -- 2. known answers: all-ones S=D=4 scale 0.5 -> ones (test_scale_factor_fix.py:138-170);
    #       S=1 -> O == V (MFAFFITests.swift:545-547) --------------------------------------
    ones = torch.ones(4, 4)
    v1 = torch.arange(16, dtype=torch.float32).reshape(1, 16) / 7.0
    np.savez_compressed(
        HERE / "known_answers.npz",
        ones_o=sdpa32(ones, ones, ones, scale=0.5),
        s1_q=torch.full((1, 16), 0.25).numpy(), s1_k=torch.full((1, 16), -0.5).numpy(),
        s1_v=v1.numpy(), s1_o=sdpa32(torch.full((1, 16), 0.25), torch.full((1, 16), -0.5), v1))

    # -- 3. conftest tensors: seed 42, randn * 0.1 (conftest.py:149-158), basic shapes
    #       (conftest.py:113-123) per dtype, dense + causal -------------------------------
    out = {}
    for (b, h, s, d) in [(1, 1, 64, 64), (1, 4, 128, 64), (1, 1, 512, 128)]:
        for dt, name in [(torch.float32, "fp32"), (torch.float16, "fp16"), (torch.bfloat16, "bf16")]:
            if (s, d) == (512, 128) and name != "bf16":
                continue
            torch.manual_seed(42)
            q = torch.randn(b, h, s, d, dtype=dt) * 0.1
            k = torch.randn(b, h, s, d, dtype=dt) * 0.1
            v = torch.randn(b, h, s, d, dtype=dt) * 0.1
            tag = f"{b}x{h}x{s}x{d}_{name}"
            out[f"q_{tag}"], out[f"k_{tag}"], out[f"v_{tag}"] = store_bits(q), store_bits(k), store_bits(v)
            out[f"o_{tag}"] = sdpa64(q, k, v)
            out[f"oc_{tag}"] = sdpa64(q, k, v, is_causal=True)
    np.savez_compressed(HERE / "conftest_shapes.npz", **out)

    # -- 4. LCG inputs, MultiHeadFFITests.swift:1228-1257,1533-1541 ("Tiny", "Small") ------
    out = {}
    for name, (b, h, s, d) in {"tiny": (1, 2, 4, 8), "small": (1, 4, 8, 16)}.items():
        n = b * h * s * d
        q = oracle.lcg_uniform(n, 12345).reshape(b, h, s, d)
        k = oracle.lcg_uniform(n, 12346).reshape(b, h, s, d)
        v = oracle.lcg_uniform(n, 12347).reshape(b, h, s, d)
        out[f"q_{name}"], out[f"k_{name}"], out[f"v_{name}"] = q, k, v
        tq, tk, tv = (torch.from_numpy(a) for a in (q, k, v))
        out[f"o_{name}"] = sdpa64(tq, tk, tv)
        out[f"oc_{name}"] = sdpa64(tq, tk, tv, is_causal=True)
    np.savez_compressed(HERE / "lcg_inputs.npz", **out)

    # -- 5. masks (mfa_prepare_mask semantics, MFABridge.swift:157-242) and Sq != Skv ------
    out = {}
    torch.manual_seed(7)
    b, h, sq, skv, d = 2, 3, 40, 72, 32
    q = torch.randn(b, h, sq, d)
    k = torch.randn(b, h, skv, d)
    v = torch.randn(b, h, skv, d)
    out["q"], out["k"], out["v"] = q.numpy(), k.numpy(), v.numpy()
    out["o_dense"] = sdpa64(q, k, v)
    out["o_causal"] = sdpa64(q, k, v, is_causal=True)  # top-left aligned, Sq != Skv
    mb = torch.rand(1, 1, sq, skv) > 0.3
    mb[..., 0] = True  # keep every row attendable
    out["mask_bool_11qk"] = mb.numpy()
    out["o_mask_bool_11qk"] = sdpa64(q, k, v, attn_mask=mb)
    mk = torch.rand(b, 1, 1, skv) > 0.5
    mk[..., 3] = True
    out["mask_bool_b11k"] = mk.numpy()
    out["o_mask_bool_b11k"] = sdpa64(q, k, v, attn_mask=mk)
    ma = torch.randn(b, h, sq, skv)
    out["mask_add_bhqk"] = ma.numpy()
    out["o_mask_add_bhqk"] = sdpa64(q, k, v, attn_mask=ma)
    m2 = torch.randn(sq, skv)  # 2-D additive, right-aligned broadcast
    out["mask_add_qk"] = m2.numpy()
    out["o_mask_add_qk"] = sdpa64(q, k, v, attn_mask=m2)
    m16 = (torch.randn(h, sq, skv)).half()
    out["mask_add_hqk_fp16"] = m16.numpy()
    out["o_mask_add_hqk_fp16"] = sdpa64(q, k, v, attn_mask=m16.float())
    np.savez_compressed(HERE / "masks.npz", **out)

    # -- 6. LSE + gradients from torch autograd in fp64 (config 3 is fwd+bwd;
    #       metal_sdpa_backend.cpp:2675-2860) ----------------------------------------------
    out = {}
    for causal in (False, True):
        torch.manual_seed(11)
        b, h, s, d = 1, 2, 96, 64
        q = torch.randn(b, h, s, d, dtype=torch.float64, requires_grad=True)
        k = torch.randn(b, h, s, d, dtype=torch.float64, requires_grad=True)
        v = torch.randn(b, h, s, d, dtype=torch.float64, requires_grad=True)
        do = torch.randn(b, h, s, d, dtype=torch.float64)
        # inputs are stored as fp32; make the fp64 leaves hold exactly the fp32 values
        with torch.no_grad():
            for t in (q, k, v):
                t.copy_(t.float().double())
            do = do.float().double()
        o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
        o.backward(do)
        sc = 1.0 / np.sqrt(d)
        scores = (q @ k.transpose(-1, -2)) * sc
        if causal:
            scores = scores.masked_fill(torch.ones(s, s).triu(1).bool(), float("-inf"))
        tag = "causal" if causal else "dense"
        out[f"q_{tag}"], out[f"k_{tag}"], out[f"v_{tag}"] = (t.detach().float().numpy() for t in (q, k, v))
        out[f"do_{tag}"] = do.float().numpy()
        out[f"o_{tag}"] = o.detach().float().numpy()
        out[f"lse_{tag}"] = torch.logsumexp(scores, -1).detach().float().numpy()
        out[f"dq_{tag}"], out[f"dk_{tag}"], out[f"dv_{tag}"] = (t.grad.float().numpy() for t in (q, k, v))
    np.savez_compressed(HERE / "backward_fp32.npz", **out)

    total = sum(p.stat().st_size for p in HERE.glob("*.npz"))
    print(f"wrote {len(list(HERE.glob('*.npz')))} fixture files, {total/1e6:.2f} MB")


def rotations() -> None:
    """Fixtures of the two rotations beside the attention path.

    RoPE: the reference's executable eager spec, `apply_rope_eager_bhsd`
    (examples/pytorch-custom-op-ffi/src/metal_sdpa_backend.cpp:1451-1468), restated here line by line in torch: fp32 math,
    interleaved pairs, out = x * cos + stack(-x_imag, x_real) * sin, cast back to x's dtype; pair-duplicated tables,
    [S, D] and [B, S, D].
    Hadamard: the contract of `mfa_hadamard_rotate` is "group-wise FWHT, normalised by 1/sqrt(N), double application =
    identity" (Sources/MFABridge/MFABridge.swift:3433-3459, AGENTS.md:161-170); the matrix itself lives in the absent
    submodule.  The fixture is y = H_N x / sqrt(N) with H_N from `scipy.linalg.hadamard` -- an INDEPENDENT Sylvester
    (natural-order) construction, explicit matrix product in fp64 -- so the oracle's butterfly and the HIP kernel are pinned
    to something neither of them computed.  (Natural vs sequency ordering cannot be recovered from the reference; natural
    is the ordering a plain FWHT butterfly produces and the one this fixture fixes.)"""
    import scipy.linalg
    out = {}
    for dt, name in [(torch.float32, "fp32"), (torch.float16, "fp16"), (torch.bfloat16, "bf16")]:
        for batched in (False, True):
            torch.manual_seed(123)
            b, h, s, d = 2, 3, 37, 64
            x = torch.randn(b, h, s, d).to(dt)
            ang = torch.rand((b, s, d // 2) if batched else (s, d // 2)) * 6.2831853
            cos_t = torch.repeat_interleave(torch.cos(ang), 2, dim=-1)
            sin_t = torch.repeat_interleave(torch.sin(ang), 2, dim=-1)
            cos_b = (cos_t.unsqueeze(0) if cos_t.dim() == 2 else cos_t).unsqueeze(1)
            sin_b = (sin_t.unsqueeze(0) if sin_t.dim() == 2 else sin_t).unsqueeze(1)
            xf = x.float()
            pairs = xf.reshape(b, h, s, d // 2, 2)
            rotated = torch.stack((-pairs[..., 1], pairs[..., 0]), -1).reshape(b, h, s, d)
            y = (xf * cos_b + rotated * sin_b).to(dt)
            tag = f"{name}_{'bsd' if batched else 'sd'}"
            out[f"x_{tag}"], out[f"cos_{tag}"], out[f"sin_{tag}"], out[f"y_{tag}"] = store_bits(x), cos_t.numpy(), sin_t.numpy(), store_bits(y)
            out[f"y32_{tag}"] = (xf * cos_b + rotated * sin_b).numpy()  # before the cast back
    np.savez_compressed(HERE / "rope.npz", **out)
    out = {}
    rng = np.random.default_rng(2024)
    for n in (2, 16, 64, 256):
        x = rng.standard_normal(3 * n).astype(np.float32)
        hm = scipy.linalg.hadamard(n).astype(np.float64) / np.sqrt(n)
        out[f"x_{n}"] = x
        out[f"y_{n}"] = (x.astype(np.float64).reshape(3, n) @ hm.T).reshape(-1)
    np.savez_compressed(HERE / "hadamard.npz", **out)
    print("wrote rope.npz, hadamard.npz")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rotations":
        rotations()  # (only the two files above: the SDPA fixtures stay byte-identical)
    else:
        main()
        rotations()
