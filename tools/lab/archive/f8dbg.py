import sys, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/universal-metal-flash-attention_amd", "/root/repo/tests"]
import umfa_torch
from oracle import oracle as orc
from tolerances import errors
def bits(t): return t.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
def run(name, B, H, Sq, Skv, vmod=None, causal=False):
    torch.manual_seed(Sq + Skv)
    q = torch.randn(B, H, Sq, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, 128, device="cuda", dtype=torch.bfloat16) * 1.7 + 0.3
    if vmod == "clamp": v = v.clamp(-6.9, 6.9)
    if vmod == "ones": v = torch.ones_like(v)
    o = umfa_torch.quantized_attention_forward_stream(q, k, v, causal=causal, quant_mode="blockwise_fp8pv")
    torch.cuda.synchronize()
    ref = orc.quantized_forward_fp8pv(bits(q), bits(k), bits(v), causal=causal)
    got = o.cpu().numpy()
    e, r = errors(got, ref)
    # per (head, q-block of 64 rows) rms
    d = ((got - ref) ** 2).reshape(B, H, Sq // 64, 64 * 128).mean(-1) / (ref ** 2).mean()
    print(name, umfa_torch.last_kernel(), "max %.3g rms %.3g" % (e, r), "worst blocks", np.sqrt(np.sort(d.ravel())[-4:]), "median block", float(np.sqrt(np.median(d))))
    if vmod == "ones": print("   ones: min/max of O", got.min(), got.max())
run("H6 S2048", 1, 6, 2048, 2048)
run("H6 S2048 clamp", 1, 6, 2048, 2048, "clamp")
run("H6 S2048 ones", 1, 6, 2048, 2048, "ones")
run("H256 S256x512 whole rounds", 1, 256, 256, 512)
run("H1 S256x2048", 1, 1, 256, 2048)
run("H2 S512x512", 1, 2, 512, 512)
run("H2 S512x512 causal", 1, 2, 512, 512, None, True)
run("H24 S4096 flux", 1, 24, 4096, 4096)
