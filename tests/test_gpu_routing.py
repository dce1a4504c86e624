"""The dispatcher's choice between the two 16-bit forward structures, pinned by measurement.

The choice is a cost model (csrc/fa_fwd16_w64.hip fwd_w64_predict_us / fwd_16_predict_us; constants from tools/fit_route_model.py over
profiles/r5/routing_random_*.jsonl).  This test replays the 49 launch sizes of the routing sweep (profiles/r5/routing_sweep_bf16.jsonl:
the sizes around which rounds 3-4 had placed their thresholds, plus the BASELINE configs) and fails when the dispatcher's kernel is more
than 10 % slower than the other one forced -- on the box the test runs on, graph-replayed, best of three."""
import json
import re
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = Path(__file__).resolve().parent.parent


def _graph_us(fn, n=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        g.replay()
        side.synchronize()
        best = 1e30
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            b.synchronize()
            best = min(best, a.elapsed_time(b) / n * 1e3)
    torch.cuda.current_stream().wait_stream(side)
    return best


def _shapes():
    out = []
    for line in open(ROOT / "profiles" / "r5" / "routing_sweep_bf16.jsonl"):
        d = json.loads(line)
        m = re.match(r"B(\d+) H(\d+) Sq(\d+) Skv(\d+) D(\d+) (\w+)", d["shape"])
        out.append(tuple(int(x) for x in m.groups()[:5]) + (m.group(6) == "causal",))
    return out + [(1, 24, 4096, 4096, 128, False), (4, 16, 1024, 1024, 64, True), (1, 4, 32768, 32768, 128, False)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_dispatcher_within_ten_percent_of_the_other_kernel(dtype):
    import umfa_torch
    behind = []
    for B, H, Sq, Skv, D, causal in _shapes():
        if dtype == torch.float16 and Sq * Skv > 4096 * 8192:
            continue
        torch.manual_seed(1)
        q = torch.randn(B, H, Sq, D, device="cuda", dtype=dtype)
        k, v = (torch.randn(B, H, Skv, D, device="cuda", dtype=dtype) for _ in range(2))
        o = torch.empty(B, H, Sq, D, device="cuda", dtype=torch.float32)
        fn = lambda: umfa_torch.attention_forward(q, k, v, causal=causal, out=o)  # noqa: E731
        t = {}
        kern = {}
        for name, opts in (("default", {}), ("w64", {"force_w64": 1}), ("r128", {"no_w64": 1})):
            with umfa_torch.options(**opts):
                t[name] = _graph_us(fn)
                kern[name] = umfa_torch.last_kernel()
        other = t["r128"] if "w64" in kern["default"] else (t["w64"] if "w64" in kern["w64"] else t["r128"])
        if t["default"] > 1.10 * other:
            behind.append((B, H, Sq, Skv, D, causal, kern["default"], round(t["default"], 1), round(other, 1)))
    assert not behind, behind
