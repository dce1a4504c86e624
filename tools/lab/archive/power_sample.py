"""Sample package power + sclk (sysfs / rocm-smi) while the forward runs back to back, for N(0,1) and all-zero operands."""
import sys, os, glob, time, threading, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "universal-metal-flash-attention_amd"))
import torch
import umfa_torch
from umfa_torch import ops

def read_power():
    out = {}
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        try: out[f.split("/")[-1]] = int(open(f).read()) / 1e6
        except Exception as e: out[f] = str(e)
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try: out["sclk"] = [l.strip() for l in open(f).read().splitlines() if "*" in l]
        except Exception as e: out["sclk"] = str(e)
    return out

B, H, S, D = 1, 24, 4096, 128
MODE = sys.argv[1] if len(sys.argv) > 1 else "fwd"   # fwd | bwd | fp8pv | int8
for name in ("randn", "zeros"):
    q, k, v = ((torch.randn if name == "randn" else torch.zeros)(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    out = torch.empty_like(q)
    if MODE == "bwd":
        o_, lse_ = ops.attention_forward(q, k, v, return_lse=True)
        do_ = torch.randn_like(q) if name == "randn" else torch.zeros_like(q)
        step = lambda: ops.attention_backward(do_, q, k, v, o_, lse_, scale=D ** -0.5)
        per = 40
    elif MODE in ("fp8pv", "int8"):
        step = lambda: umfa_torch.quantized_attention_forward_stream(q, k, v, quant_mode="blockwise_fp8pv" if MODE == "fp8pv" else "blockwise")
        per = 200
    else:
        step = lambda: ops.attention_forward(q, k, v, out=out)
        per = 200
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(per): step()
    stop = False
    def work():
        while not stop:
            g.replay(); torch.cuda.synchronize()
    t = threading.Thread(target=work); t.start()
    time.sleep(1.0)
    for i in range(6):
        smi = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        lines = [l.split(":", 1)[1].strip() for l in smi.splitlines() if "Power (W)" in l or "sclk" in l]
        print(name, read_power(), lines, flush=True)
        time.sleep(0.4)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stop = True; t.join()
    a.record(); g.replay(); b.record(); b.synchronize()
    print(MODE, name, "ms per step in the graph:", a.elapsed_time(b) / per, flush=True)
