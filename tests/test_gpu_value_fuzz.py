"""GPU: adversarial VALUE fuzz of the forward kernels (tools/lab/value_fuzz.py: score shifts of hundreds of nats in both
directions, large / tiny score scales, attention sinks, sign flips from key tile to key tile, zero rows) through every forward
kernel family -- w64 bf16 / fp16 at head_dim 128 and 64, causal, sliding window, the 128-row kernel with and without masks, the
int8 kernel -- against an fp64 restatement (16-bit kernels: max error <= 1.5 ulp of P at 1.0) or the oracle's quantised forward
(int8: 2.5e-3); further legs: arbitrary shapes through the forced w64 families, the backward, the runtime-quantised forward and
backward, grouped K / V heads, fused RoPE, mask tensors, the blocking host-buffer ABI, concurrent streams and hipGraph capture.
Found in round 3 by these sweeps: lazy-mode underflow of rows that start a segment on the reference 0; a window whose left
extent was clamped to Skv instead of Sq; fused RoPE at head_dim 64 failing with error 5; the split-KV fold silently missing from
the second replay of a captured graph on."""
import importlib.util
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = Path(__file__).resolve().parent.parent


def _fuzz():
    # loaded ONCE: the module puts its directories in front of sys.path when it runs, and a sys.path that has grown by three entries per
    # test becomes a PYTHONPATH too long to exec with (torch.compile's worker pool in a later test: "Argument list too long")
    if "value_fuzz" not in sys.modules:
        spec = importlib.util.spec_from_file_location("value_fuzz", ROOT / "tools" / "lab" / "value_fuzz.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules["value_fuzz"] = mod
        spec.loader.exec_module(mod)
    return sys.modules["value_fuzz"]


@pytest.mark.parametrize("seed", range(150))
def test_forward_adversarial_values(seed):
    msg = _fuzz().run_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(60))
def test_backward_adversarial_values(seed):
    msg = _fuzz().run_bwd_case(seed)
    assert msg is None, msg


def test_backward_is_a_function_of_the_o_it_is_given():
    """seed 401643 of the round-6 soak ('sink_mid' + causal, bf16, B1 H3 S256 D64): dQ 0.34 of the largest gradient away from fp64 autograd on two rows -- and the
    fp32-EXACT engine lands on the same numbers (tools/lab/bwd_sink_mid_probe.py, profiles/r6/bwd_sink_mid_probe.txt): P is one-hot on a key 57 x the median there,
    dS = P (dP - D) cancels, and D = rowsum(dO O) is taken from the forward's 16-bit O, as in any backward that autograd hands a saved 16-bit output.  Against fp64
    gradients of the same function WITH that O the kernels are inside the leg's tolerance; the exact engine and the 16-bit engine agree with each other."""
    import umfa_torch
    fz = _fuzz()
    assert fz.run_bwd_case(401643) is None
    import random
    rng = random.Random(401643 + 100000)
    kind = rng.choice([k_ for k_ in fz.KINDS if k_ != "zero_rows"])
    assert kind == "sink_mid"
    g = torch.Generator(device="cuda").manual_seed(401643)
    q, k, v, do = (torch.randn(1, 3, 256, 64, device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(4))
    for _ in range(5):
        rng.random()  # (the leg's draws of dtype, head_dim, heads, lengths, causal: fixed above)
    q, k, v = fz.transform(random.Random(1), q, k, v, "sink_mid", {})
    grads = {}
    for name, opts in (("bwd16", {}), ("exact", {"bwd_exact": 1})):
        qg, kg, vg = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        with umfa_torch.options(**opts):
            umfa_torch.scaled_dot_product_attention(qg, kg, vg, is_causal=True).backward(do)
        grads[name] = qg.grad.float()
    assert float((grads["bwd16"] - grads["exact"]).abs().max()) <= fz.GTOL[torch.bfloat16] * float(grads["exact"].abs().max())


@pytest.mark.parametrize("seed", range(120))
def test_w64_families_random_shapes(seed):
    msg = _fuzz().run_shape_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(80))
def test_quantized_forward_random_cases(seed):
    msg = _fuzz().run_i8_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(60))
def test_gqa_random_cases(seed):
    msg = _fuzz().run_gqa_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(60))
def test_fused_rope_random_cases(seed):
    msg = _fuzz().run_rope_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(40))
def test_concurrent_streams_random_cases(seed):
    msg = _fuzz().run_streams_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(30))
def test_graph_capture_random_cases(seed):
    msg = _fuzz().run_graph_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(100))
def test_mask_tensor_random_cases(seed):
    msg = _fuzz().run_mask_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(60))
def test_blocking_host_abi_random_cases(seed):
    msg = _fuzz().run_host_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(40))
def test_quantized_forward_backward_random_cases(seed):
    msg = _fuzz().run_qbwd_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(40))
def test_prequantized_backward_random_cases(seed):
    msg = _fuzz().run_prequant_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(40))
def test_rotations_random_cases(seed):
    msg = _fuzz().run_aux_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(20))
def test_host_threads_random_cases(seed):
    msg = _fuzz().run_threads_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(24))
def test_launch_size_random_cases(seed):
    msg = _fuzz().run_big_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(60))
def test_backward_random_shapes(seed):
    msg = _fuzz().run_bwd_shape_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(40))
def test_wide_head_dims_forward_and_backward(seed):
    msg = _fuzz().run_wide_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(60))
def test_quantized_forward_with_random_caller_masks(seed):
    msg = _fuzz().run_qmask_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(120))
def test_w64_mask_kernels_random_cases(seed):
    """(round 6) additive fp16 / bf16 mask tensors on fa_fwd16_w64<., 128, bias> and bool tensors on the int8 kernel's mask instantiation, forced, random whole-tile
    shapes and forced small grids (cut blocks, several segments per workgroup -- the regime in which the additive kernels' first build was wrong)"""
    msg = _fuzz().run_w64_mask_case(seed)
    assert msg is None, msg



@pytest.mark.parametrize("seed", range(120))
def test_paired_causal_schedule_random_cases(seed):
    """the 128-row kernel's balanced causal pairs (round 6, option cbal): random launches, every cut, ragged lengths, V outside fp16's range, graph replays"""
    msg = _fuzz().run_cbal_case(seed)
    assert msg is None, msg


@pytest.mark.parametrize("seed", range(120))
def test_decode_like_launches_random_cases(seed):
    """1 ... 32 query rows: the decode form and the plain form of the 128-row kernel, forced split-KV part counts (round 6: the fold's 16-byte slots and read-ahead)"""
    msg = _fuzz().run_decode_case(seed)
    assert msg is None, msg
