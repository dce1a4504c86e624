"""Build-time contracts of the 64-rows-per-wave kernels (fa_fwd16_w64.hip), checked without a GPU:

1. Register ownership.  The kernels address v[128:255] (score tiles) and a[128:255] (O^T) by literal number inside inline
   asm and rely on `amdgpu_num_vgpr(128)` keeping the COMPILER inside v[0:127] + a[0:127] (the lab build with matrix-pipe
   row sums, W64_MSUM_ON, also owns v[118:127] and is compiled with 118).  The test compiles the file to
   gfx950 assembly and checks that no instruction outside the asm blocks names a register above 127, that every kernel
   still gets the full 512-register file, and that nothing is spilled to scratch inside the tile loop.
2. The committed generated instruction streams (*_body.inc, *_regs.inc) are exactly what tools/gen_w64_body.py emits now
   (its self-check - every consumer behind its producer - runs as part of the generation)."""
import os
import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "universal-metal-flash-attention_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    text = ""
    for src in ("fa_fwd16_w64.hip", "fa_fwd16_w64_bias.hip"):  # (the additive-mask families are a translation unit of their own since round 6)
        out = tmp_path_factory.mktemp("w64") / (src + ".s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-w", "--cuda-device-only",
                               "-S", str(CSRC / src), "-o", str(out)], cwd=CSRC)
        text += out.read_text()
    return text


def _kernels(text):
    """name -> list of lines of the kernel body"""
    out, cur, name = {}, None, None
    for line in text.split("\n"):
        m = re.match(r"^(_ZN4umfa\w+):", line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
        elif cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                cur = None
    return out


def test_compiler_stays_in_the_lower_register_halves(asm):
    kernels = _kernels(asm)
    # bf16 / fp16 x {fp32, 16-bit O} x {causal, not} + bf16 / fp16 x {causal, not} with the fused Q rotation (16-bit O)
    # + int8 x {causal, not} + int8-fp8 x {causal, not}
    # + the head_dim 64 family: bf16 / fp16 x {fp32, 16-bit O} x {causal, not}
    # + sliding-window instantiations at head_dim 128: bf16 / fp16 x {fp32, 16-bit O}
    # + bf16 Q / K with fp16 P V (option pv_fp16): {fp32, bf16 O} x {causal, not}
    # + sliding-window instantiations at head_dim 64: bf16 / fp16 x {fp32, 16-bit O}
    # + pv_fp16 at head_dim 64: {fp32, bf16 O} x {causal, not}
    # + (round 4: pv_fp16 is the default bf16 arithmetic) its fused-rotation {causal, not} and window
    #   {fp32, bf16 O} forms at head_dim 128 and its window forms at head_dim 64
    # + bool mask tensors (MASKT): bf16-pv16 / fp16 x {fp32, 16-bit O} at head_dim 128
    # + (round 5) the same four at head_dim 64
    # + (round 6) additive fp16 mask tensors (MASKA, fa_fwd16_w64_bias.hip): bf16-pv16 / fp16 x {fp32, 16-bit O} at head_dim 128
    # + (round 6) the int8 kernel's bool-mask instantiation (fp32 O); the additive-mask kernels at head_dim 64 (four more)
    assert len(kernels) == 63, sorted(kernels)
    for name, lines in kernels.items():
        in_asm, vmax, amax, n_mfma, loop_scratch = False, 0, 0, 0, 0
        if "w64_bias" in name:
            # NO scratch anywhere in the additive-mask kernels: their first woven build spilled three lane constants of the fp32-output epilogue (reloads in
            # the segment loop, none in the tile loop) and returned WRONG results from a workgroup's second segment on -- found on the GPU, not by a
            # compiler diagnostic (the kernel text keeps those constants out of kernel scope now: `tid_s`, fa_fwd16_w64_kernel.inc)
            assert not any("scratch_" in l for l in lines), name
        mfma_seen = 0
        depth = 0  # loop depth of the current basic block, from the label comments ("in Loop: Header=... Depth=N")
        total_mfma = sum("v_mfma" in l for l in lines)
        for l in lines:
            if re.match(r"^\.LBB\d+_\d+:", l):
                m = re.search(r"Depth=(\d+)", l)
                depth = int(m.group(1)) if m else 0
            if "ASMSTART" in l:
                in_asm = True
            elif "ASMEND" in l:
                in_asm = False
            elif in_asm:
                mfma_seen += "v_mfma" in l
            else:
                for m in re.findall(r"\bv(\d+)\b", l):
                    vmax = max(vmax, int(m))
                for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", l):
                    vmax = max(vmax, int(b))
                for m in re.findall(r"\ba(\d+)\b", l):
                    amax = max(amax, int(m))
                for a, b in re.findall(r"\ba\[(\d+):(\d+)\]", l):
                    amax = max(amax, int(b))
                # the tile loop is the depth-2 loop (depth 1 = the workgroup's segments: prologue, drain, output, fold)
                if "scratch_" in l and depth >= 2:
                    loop_scratch += 1
        assert total_mfma >= 150, (name, total_mfma)          # 8 tile-body variants of 16-64 inline-asm MFMAs each
        # the int8 kernels own literal registers below 128 as well: fp8 variant v[96:127] / a[96:127] (amdgpu_num_vgpr(96)),
        # int8 + fp16 variant its bias tile v[112:127] (amdgpu_num_vgpr(112))
        lim = 95 if "i8f8" in name else 111 if "w64_i8" in name else 127
        assert vmax <= lim and amax <= lim, (name, vmax, amax)  # the compiler never names our registers
        assert loop_scratch == 0, (name, loop_scratch)          # no scratch traffic inside the tile loop


def test_every_kernel_gets_512_registers(asm):
    nxt = [int(x) for x in re.findall(r"\.amdhsa_next_free_vgpr (\d+)", asm)]
    acc = [int(x) for x in re.findall(r"\.amdhsa_accum_offset (\d+)", asm)]
    # the hardware allocates in granules of 8 registers: 511 (clobbers name v254 / a254, the highest names hipcc does not reserve) is 512
    assert len(nxt) == 63 and all((n + 7) // 8 * 8 == 512 for n in nxt), nxt
    assert all(a == 256 for a in acc), acc


def test_generated_streams_are_current(tmp_path):
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("W64_"):
            del env[k]
    env["W64_OUT"] = str(tmp_path / "b16.inc")
    env["W64_OUT_I8"] = str(tmp_path / "bi8.inc")
    env["W64_OUT_I8F8"] = str(tmp_path / "bi8f8.inc")
    env["W64_OUT_D64"] = str(tmp_path / "bd64.inc")
    env["W64_OUT_BIAS"] = str(tmp_path / "bbias.inc")
    env["W64_OUT_BIAS_D64"] = str(tmp_path / "bbiasd64.inc")
    regs = (CSRC / "fa_fwd16_w64_regs.inc").read_text()
    subprocess.check_call([sys.executable, str(ROOT / "tools" / "gen_w64_body.py")], env=env, stdout=subprocess.DEVNULL)
    assert (tmp_path / "b16.inc").read_text() == (CSRC / "fa_fwd16_w64_body.inc").read_text()
    assert (tmp_path / "bd64.inc").read_text() == (CSRC / "fa_fwd16_w64d64_body.inc").read_text()
    assert (tmp_path / "bi8.inc").read_text() == (CSRC / "fa_fwd_w64_i8_body.inc").read_text()
    assert (tmp_path / "bi8f8.inc").read_text() == (CSRC / "fa_fwd_w64_i8f8_body.inc").read_text()
    assert (tmp_path / "bbias.inc").read_text() == (CSRC / "fa_fwd16_w64_bias_body.inc").read_text()
    assert (tmp_path / "bbiasd64.inc").read_text() == (CSRC / "fa_fwd16_w64d64_bias_body.inc").read_text()
    assert (CSRC / "fa_fwd16_w64_regs.inc").read_text() == regs  # the helper file is rewritten in place: unchanged


def test_lds_images_are_conflict_free_in_the_bank_model():
    """tools/lds_bank_check.py models the LDS bank rules for every tile image the kernels use (forward K / V images,
    the dual-use row + transposed images of the bf16 backward at head_dim 64 / 128 / 256): all must read 1-way."""
    import subprocess
    import sys
    out = subprocess.check_output([sys.executable, str(ROOT / "tools" / "lds_bank_check.py")], text=True)
    lines = [ln for ln in out.splitlines() if "-way" in ln]
    assert len(lines) >= 7, out
    for ln in lines:
        for tok in ln.replace(",", " ").split():
            if tok.endswith("-way"):
                assert tok == "1-way", ln
