#!/usr/bin/env python3
"""Measured forward rel-err of every BASELINE config at FULL size against the CPU oracle (row subset, all keys), and
the operand-format floor beside it (oracle.flash_format_floor).  python tools/parity_probe.py [--full] > json"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import umfa_torch  # noqa: E402
from oracle import oracle, parity  # noqa: E402

FULL = "--full" in sys.argv
res = {}


def run(name, B, H, S, D, dt, causal=False, out_dtype=None, seed=0, scale_in=1.0):
    torch.manual_seed(seed)
    q, k, v = ((torch.randn(B, H, S, D, device="cuda", dtype=torch.float32) * scale_in).to(dt) for _ in range(3))
    o = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=out_dtype or dt)
    torch.cuda.synchronize()
    rows = np.arange(S) if FULL and B * H * S * S * D * 4.0 < 6e11 else None
    r = parity.forward_rel_err(q, k, v, o, causal=causal, rows=rows, floor_kind={torch.bfloat16: "bf16", torch.float16: "fp16"}[dt])
    r["kernel"] = umfa_torch.last_kernel()
    res[name] = r
    print(name, r, file=sys.stderr, flush=True)


bf, hf, f32 = torch.bfloat16, torch.float16, torch.float32
run("cfg2_B4_H16_S1024_D64_bf16_causal_fp32O", 4, 16, 1024, 64, bf, True, f32)
run("cfg2_B4_H16_S1024_D64_bf16_causal_bf16O", 4, 16, 1024, 64, bf, True, bf)
run("cfg3_flux_bf16_fp32O", 1, 24, 4096, 128, bf, False, f32)
run("cfg3_flux_bf16_bf16O", 1, 24, 4096, 128, bf, False, bf)
run("cfg3_flux_bf16_causal_fp32O", 1, 24, 4096, 128, bf, True, f32)
run("cfg3_flux_fp16_fp32O", 1, 24, 4096, 128, hf, False, f32)
run("cfg3_flux_fp16_fp16O", 1, 24, 4096, 128, hf, False, hf)
run("cfg3_flux_bf16_fp32O_x0.1", 1, 24, 4096, 128, bf, False, f32, seed=42, scale_in=0.1)
run("cfg5_shard_B1_H4_S32768_D128_bf16_fp32O", 1, 4, 32768, 128, bf, False, f32)
run("cfg5_shard_B1_H4_S32768_D128_bf16_bf16O", 1, 4, 32768, 128, bf, False, bf)
run("s256_H256_bf16_fp32O", 1, 256, 256, 128, bf, False, f32)
run("s8192_H16_bf16_fp32O", 1, 16, 8192, 128, bf, False, f32)

# cfg4: int8 block-wise vs oracle.quantized_forward on a few heads (the oracle quantises whole slabs: head subset, all rows)
torch.manual_seed(0)
B, H, S, D = 1, 16, 8192, 128
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.float32).to(bf) for _ in range(3))
for bits_ in (8, 4):
    o8 = umfa_torch.quantized_attention_forward_stream(q, k, v, bits=bits_)
    torch.cuda.synchronize()
    hs = [0, 7, 15]
    sub = lambda t: parity.bits(t[:, hs])  # noqa: E731
    rows = parity.sample_rows(S)
    qs, ks, vs = sub(q), sub(k), sub(v)
    # fake-quantise with the oracle's quantiser (whole slabs), then the oracle on the row subset
    def fq(x):
        f = oracle.to_f32(x).reshape(len(hs), S * D)
        out = np.empty_like(f)
        for i in range(len(hs)):
            qi, sc = oracle.quantize_symmetric(f[i], group=64 * D, bits=bits_)
            out[i] = oracle.dequantize(qi, sc, group=64 * D)
        return out.reshape(1, len(hs), S, D)
    ref_q = oracle.sdpa_forward_rows(fq(qs), fq(ks), fq(vs), rows)
    ref_x = oracle.sdpa_forward_rows(qs, ks, vs, rows)
    got = o8[:, hs].float().cpu().numpy()[:, :, rows]
    res[f"cfg4_int{bits_}_blockwise_B1_H16_S8192"] = {"rel_vs_quantized_oracle": parity.rel_err(got, ref_q),
                                                     "rel_vs_exact_sdpa": parity.rel_err(got, ref_x),
                                                     "oracle_quantized_vs_exact": parity.rel_err(ref_q, ref_x),
                                                     "kernel": umfa_torch.last_kernel(), "heads": hs, "rows": int(rows.size)}
    print(res[f"cfg4_int{bits_}_blockwise_B1_H16_S8192"], file=sys.stderr, flush=True)
print(json.dumps(res))
