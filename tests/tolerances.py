"""Forward parity bounds of the 16-bit kernels -- derived from the operand FORMAT, not fitted to a kernel.

Metric: max = max|O - O_ref| / max|O_ref|, rms = rms(O - O_ref) / rms(O_ref), O_ref = the fp64
CPU oracle on the already rounded inputs, fp32 O at the ABI.

Two assertions, both independent of what any kernel measured last week:

1. FLOOR-RELATIVE (whenever the call site hands over the inputs).  `oracle.flash_format_floor` is what an IDEAL flash
   kernel gives on the same rows: exact fp64 scores, exponentials and sums, P rounded ONCE to the operand type because
   the P V MFMA takes nothing wider.  Its distance from the oracle is the part of the error the format fixes
   (profiles/r2/error_anatomy.md: bf16 0.8e-3 (S = 256) ... 1.6e-3 (S >= 4096) max, ~1.5e-3 rms; fp16 8x smaller).
   A kernel is held to a multiple of THAT, on the same rows:
     exact running max (fa_fwd16; fa_fwd16_w64 with softmax_reference = exact)      max <= 1.15 x floor, rms <= 1.05 x floor
     stale reference (fa_fwd16_w64 lazy / deferred max: a row's largest P is no      max <= 2.5 x floor,  rms <= 1.15 x floor
       longer exactly 1.0, so the dominant key takes the ordinary half-ulp too)
   The rms bound is the sharp one: a kernel that got 15 % worse everywhere fails it in every regime.
2. FORMAT CEILING (every call, also without inputs: masks, windows, fuzz shapes): max <= one ulp of P at 1.0
   (2^-8 bf16, 2^-11 fp16), rms <= 0.6 of it.  The reference's own tolerances are 1e-2 (bf16) / 1e-3 (fp16)
   (examples/pytorch-custom-op-ffi/tests/conftest.py:186-199).

The north-star's 1e-3 is ASSERTED for every kernel whose P V product runs in fp16: the fp16 kernels and -- the default since
round 4 -- the bf16-input kernels ("pv16" in the kernel name: bf16 Q K^T, P rounded to fp16, V converted bf16 -> fp16 inside
the kernel).  The format that matters for the bounds is P's, so those kernels are held to fp16's floor and ceiling.  Only the
bf16 P V kernels (option pv_fp16 = 0, or the fallback after a status word was raised) keep bf16's floor: it is above 1e-3 from
S ~ 1024 on.
"""
from __future__ import annotations

import json
import os

import numpy as np

ULP_AT_ONE = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}
# regime -> (multiple of the floor's max, multiple of the floor's rms)
FLOOR_MULT = {"exact": (1.15, 1.05), "stale": (2.5, 1.15)}  # measured worst (profiles/r3/parity_record.jsonl): 1.00 / 1.00 and 1.38 / 1.09
NORTH_STAR = 1.0e-3
MAX_FLOOR_ROWS = 64  # the floor emulation holds [B, H, rows, Skv] in fp64
# the MAX of a few thousand elements is a noisy statistic (kernel and ideal kernel round different P at the binade
# edges); below this many compared elements the max multiple is widened by SMALL_SAMPLE_SLACK, the rms multiple is not
SMALL_SAMPLE, SMALL_SAMPLE_SLACK = 1 << 17, 1.25  # measured worst small-sample max ratio: 1.08 (exact), 1.19 (stale)


def _name(dt) -> str:
    s = str(dt)
    return "fp16" if "float16" in s and "bfloat16" not in s or s == "fp16" else "bf16" if "bf" in s else s


def errors(o, ref):
    o = np.asarray(o, np.float64)
    ref = np.asarray(ref, np.float64)
    d = o - ref
    return (float(np.abs(d).max() / max(np.abs(ref).max(), 1e-30)),
            float(np.sqrt((d * d).mean() / max((ref * ref).mean(), 1e-60))))


def record(tag: str, **vals) -> None:
    path = os.environ.get("UMFA_PARITY_RECORD")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"tag": tag, "test": os.environ.get("PYTEST_CURRENT_TEST", ""), **vals}) + "\n")


def fam(kernel: str) -> str:
    """kernel family: the name without the ",pv16" tag (bf16 operands, P V product in fp16 -- the default bf16 arithmetic) and without
    ",ks2" (the key-split form of the 128-row head_dim-64 kernel: same arithmetic, eight waves per workgroup)"""
    return kernel.replace(",pv16", "").replace(",ks2", "").replace(",pipe", "")


def regime_of(kernel: str) -> str:
    """"exact": the kernel's softmax reference is the exact running max; "stale": fa_fwd16_w64 in its lazy / deferred modes and, since
    round 4, the 128-row fa_fwd16 (deferred reference, tau = 6: fa_fwd_16_kernel.h)."""
    if kernel.startswith("fa_fwd16<"):
        return "stale"
    if not kernel.startswith("fa_fwd16_w64"):
        return "exact"
    try:  # the LIBRARY's live values (they may have been seeded from the environment: UMFA_W64_TAU, UMFA_W64_LAZY)
        from umfa_torch import ops
        mode = ops.get_option("softmax_reference")
        tau = float(ops.get_option("softmax_tau"))
    except Exception:  # noqa: BLE001  (no device / an older library: its defaults)
        mode, tau = "default", 6.0
    return "exact" if mode == "exact" or (mode == "deferred" and tau == 0.0) else "stale"


# a 16-bit OUTPUT (fused cast-back epilogue) adds one rounding of O itself: at most half an ulp of the value (2^-8 of it
# for bf16, 2^-11 for fp16) in max, about 0.41 of that in rms (uniform error, values log-uniform inside a binade)
OUT_HALF_ULP = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}


def check_forward(o, ref, dt, kernel: str, tag: str = "", scale_max: float = 1.0, min_elems_for_rms: int = 4096,
                  out_dt=None, inputs=None, rows=None, causal: bool = False, scale=None):
    """Assert the forward parity bounds for one output against the oracle.

    o, ref: [B, H, R, D] (R = all query rows, or the subset `rows` of them).  kernel: umfa_torch.last_kernel() /
    ctx.last_kernel.  inputs = (q, k, v) as the oracle takes them (bf16 as uint16 bits): enables the floor-relative
    assertion on (a spread of <= 64 of) the same rows; rows / causal / scale describe them.  scale_max loosens the max
    bounds for deliberately hostile inputs (stated at the call site); out_dt: the element type O was stored in when it is
    not fp32."""
    name = _name(dt)
    if name == "bf16" and "pv16" in kernel:
        name = "fp16"  # the format of P (and of V inside the kernel) decides the bounds
    regime = regime_of(kernel)
    mx, rms = errors(o, ref)
    # rms: 0.6 ulp -- a row that spreads over very many keys (S = 131072, flat) sits at the rounding level of P itself, ulp x 0.41 ...
    # 0.5 (measured with fp16 P: 0.503 ulp); the sharp rms bound is the floor-relative one below
    cmax, crms = ULP_AT_ONE[name], 0.6 * ULP_AT_ONE[name]
    if out_dt is not None and _name(out_dt) in OUT_HALF_ULP:
        h = OUT_HALF_ULP[_name(out_dt)]
        # max: the kernel's error and the output rounding can meet in one element, and an element just above a power of two is off
        # by up to h of ITSELF (measured at config 5's shard, bf16 O: 3.04e-3 of max|O|); rms: independent, uniform inside a binade
        cmax, crms = cmax + h, float(np.hypot(crms, 0.5 * h))
    rec = dict(dtype=name, kernel=kernel, regime=regime, out=_name(out_dt) if out_dt is not None else "fp32", max=mx, rms=rms,
               ceiling_max=cmax * scale_max, ceiling_rms=crms * scale_max, n=int(np.asarray(ref).size))
    fl = None
    if inputs is not None and out_dt is None:
        from oracle import oracle
        qb, kb, vb = inputs
        R = np.asarray(ref).shape[2]
        grows = np.arange(R) if rows is None else np.asarray(rows)
        pos = np.unique(np.linspace(0, R - 1, min(R, MAX_FLOOR_ROWS)).astype(np.int64))
        floor_o = oracle.flash_format_floor(qb, kb, vb, grows[pos], name, scale=scale, causal=causal)
        fmax, frms = errors(floor_o, np.asarray(ref)[:, :, pos])
        kmax, krms = errors(np.asarray(o)[:, :, pos], np.asarray(ref)[:, :, pos])
        fl = (fmax, frms, kmax, krms)
        rec.update(floor_max=fmax, floor_rms=frms, max_on_floor_rows=kmax, rms_on_floor_rows=krms)
    record(tag or kernel, **rec)
    assert mx < cmax * scale_max, (tag, kernel, "format ceiling, max", mx, cmax * scale_max)
    if name == "fp16" and scale_max == 1.0 and out_dt is None:  # (a 16-bit O adds its own rounding: up to 2^-8 of an element for bf16)
        assert mx <= NORTH_STAR, (tag, kernel, "north-star 1e-3", mx)  # fp16 P V (fp16 inputs, or bf16 inputs by default) meets the stated tolerance everywhere
    if np.asarray(ref).size >= min_elems_for_rms:
        assert rms < crms * scale_max, (tag, kernel, "format ceiling, rms", rms, crms * scale_max)
    if fl is not None:
        fmax, frms, kmax, krms = fl
        mmax, mrms = FLOOR_MULT[regime]
        if np.asarray(ref)[:, :, :MAX_FLOOR_ROWS].size < SMALL_SAMPLE:
            mmax *= SMALL_SAMPLE_SLACK
        assert kmax <= mmax * scale_max * fmax, (tag, kernel, regime, "max vs format floor", kmax, fmax, mmax * scale_max)
        if np.asarray(ref)[:, :, :MAX_FLOOR_ROWS].size >= min_elems_for_rms:
            assert krms <= mrms * scale_max * frms, (tag, kernel, regime, "rms vs format floor", krms, frms, mrms * scale_max)
    return mx, rms
