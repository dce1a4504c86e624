"""Where umfa.flash_attention_forward's time goes at the FLUX shape (host arrays): output allocation, wrapping (mfa_buffer_from_ptr: an HBM
mirror per array), the C call, destroying the wrappers."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import umfa  # noqa: E402
from umfa import core  # noqa: E402

import umfa_torch  # noqa: E402

for shape, chunks in (((1, 24, 4096, 128), 1), ((1, 24, 4096, 128), 0), ((1, 24, 4096, 128), 1), ((1, 24, 4096, 128), 0)):
    umfa_torch.set_option("sync_chunks", chunks)
    B, H, S, D = shape
    rng = np.random.default_rng(0)
    q, k, v = ((rng.standard_normal(shape, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16) for _ in range(3))
    rows = []
    with umfa.MFAContext() as ctx:
        for it in range(4):
            t = [time.perf_counter()]
            o = np.zeros(shape, np.float32); t.append(time.perf_counter())
            bufs = [core.MFABuffer(ctx, a) for a in (q, k, v, o)]; t.append(time.perf_counter())
            core._check_error(core._lib.mfa_attention_forward(ctx.handle, *(x.handle for x in bufs), B, S, S, H, D, float(D) ** -0.5, False, core.MFA_PRECISION_BF16,
                                                              core.MFA_PRECISION_BF16, core.MFA_PRECISION_FP32, False, False, False, False, None, 0, None, None, 0,
                                                              core.MFA_MASK_TYPE_NONE, core.MFA_MASK_SCALAR_BYTE)); t.append(time.perf_counter())
            per = []
            for x in bufs:
                t0 = time.perf_counter()
                x.close()
                per.append(round((time.perf_counter() - t0) * 1e3, 3))
            t.append(time.perf_counter())
            rows.append([round((b - a) * 1e3, 3) for a, b in zip(t, t[1:])] + [per])
    print(json.dumps({"shape": shape, "sync_chunks": chunks, "ms [np.zeros, wrap x4, call, destroy x4] per iteration": rows}))
