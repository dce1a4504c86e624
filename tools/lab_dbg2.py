import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
torch.manual_seed(0)
causal = sys.argv[1] == "1"
for (Sq, Skv) in [(256, 64), (256, 128), (256, 192), (256, 256), (512, 512)]:
    q = torch.randn(1, 1, Sq, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(1, 1, Skv, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(1, 1, Skv, 128, device="cuda", dtype=torch.bfloat16)
    od = torch.bfloat16 if len(sys.argv) > 2 else torch.float32
    o = umfa_torch.attention_forward(q, k, v, causal=causal, out_dtype=od)[0, 0].float()
    if causal:
        ii = torch.arange(Sq, device="cuda")[:, None]; jj = torch.arange(Skv, device="cuda")[None, :]
        m = torch.zeros(Sq, Skv, device="cuda"); m[jj > ii] = float("-inf")
    else:
        m = 0
    ref = torch.softmax((q.float()[0, 0] @ k.float()[0, 0].T) * 128 ** -0.5 + m, -1) @ v.float()[0, 0]
    err = (o - ref).abs().amax(-1)
    bad = (~(err < 0.02)).nonzero().flatten().tolist()
    if causal and "FULLREF" in __import__("os").environ:
        ref2 = torch.softmax((q.float()[0, 0] @ k.float()[0, 0].T) * 128 ** -0.5, -1) @ v.float()[0, 0]
        bad = (~((o - ref2).abs().amax(-1) < 0.02)).nonzero().flatten().tolist()
    print(Sq, Skv, "causal" if causal else "full", umfa_torch.last_kernel(), "bad rows", len(bad), bad[:12])
