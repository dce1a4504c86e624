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
