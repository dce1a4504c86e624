"""Build-time contract of the pinned backward pipelines (fa_bwd_16.hip, head_dim 128), checked without a GPU.

The S / dP accumulators of bwd16_dkdv and bwd16_dq2 are written by inline-asm MFMAs (VGPR destination, Mma16::mma_v) and
read by vector instructions a fixed number of MFMA issues later.  That only holds while the tile loops are FULLY
unrolled: a rolled loop indexes the accumulator arrays dynamically, i.e. through scratch, and stores them right behind
the MFMA that has not written them yet (seen in round 2: deterministic garbage in one causal instantiation).  So: every
head_dim-128 kernel of the file is scratch-free, contains the expected number of MFMAs, and its steady tile loop carries
no v_accvgpr copies."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "universal-metal-flash-attention_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    mk = (CSRC / "Makefile").read_text()
    m = re.search(r"build/fa_bwd_16\.o: EXTRA \+= (.*)", mk)
    assert m and "-pragma-unroll-threshold" in m.group(1) and "-fno-slp-vectorize" in m.group(1), "fa_bwd_16.o lost its flags"
    out = tmp_path_factory.mktemp("bwd16") / "fa_bwd_16.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-DBWD16_LAB_ONLY128", "--cuda-device-only", "-S",
                           *m.group(1).split(), str(CSRC / "fa_bwd_16.hip"), "-o", str(out)], cwd=CSRC)
    return out.read_text()


def _kernels(text):
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        name, meta = m.group(1), m.group(2)
        k0 = text.index(name + ":")
        body = text[k0:text.index("s_endpgm", k0)]
        get = lambda key: int(re.search(key + r":\s+(\d+)", meta).group(1))  # noqa: E731
        out[name] = dict(body=body, scratch=get(r"\.private_segment_fixed_size"), spill=get(r"\.vgpr_spill_count"))
    return out


def test_pinned_pipelines_are_unrolled_and_scratch_free(asm):
    ks = _kernels(asm)
    want = {"bwd16_dkdv_kernel": 4 * 64, "bwd16_dq2_kernel": 3 * 48, "bwd16_dq_kernel": 3 * 24}  # tile bodies x MFMAs per tile
    seen = 0
    for name, k in ks.items():
        for key, n_mfma in want.items():
            if key in name:
                seen += 1
                assert k["scratch"] == 0 and k["spill"] == 0, (name, k["scratch"], k["spill"])
                assert k["body"].count("v_mfma") == n_mfma, (name, k["body"].count("v_mfma"), n_mfma)
    # {dkdv, dq, dq2} x {causal, not} + the dS-storing dkdv of the lab option bwd_ds_store, bf16 (the lab switch compiles bf16
    # head_dim 128 only)
    assert seen == 7, sorted(ks)
    gemm = [k for n, k in ks.items() if "bwd16_dq_gemm_kernel" in n]
    assert len(gemm) == 1 and gemm[0]["scratch"] == 0 and gemm[0]["spill"] == 0 and gemm[0]["body"].count("v_mfma") == 16


def test_steady_loops_have_no_accvgpr_copies(asm):
    for name, k in _kernels(asm).items():
        if "bwd16_dkdv_kernel" not in name and "bwd16_dq2_kernel" not in name:
            continue
        lines = k["body"].split("\n")
        idx = [i for i, l in enumerate(lines) if "v_mfma" in l]
        per_pair = 128 if "dkdv" in name else 96
        best = min(range(len(idx) - per_pair + 1), key=lambda q: idx[q + per_pair - 1] - idx[q])  # the densest run = the unrolled steady pair
        seg = lines[idx[best]:idx[best + per_pair - 1] + 1]
        copies = sum(("v_accvgpr_read" in l) or ("v_accvgpr_write" in l) or ("v_accvgpr_mov" in l) for l in seg)
        assert copies == 0, (name, copies)
