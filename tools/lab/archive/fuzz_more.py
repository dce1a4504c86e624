"""Extended run of the GPU fuzz cases (tests/test_gpu_fuzz.py) over seeds the suite does not contain."""
import sys, os, traceback
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "universal-metal-flash-attention_amd"), os.path.join(ROOT, "tests")]
import test_gpu_fuzz as F
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    for fn in (F.test_forward_random_case, F.test_backward_random_case):
        try:
            fn(seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", fn.__name__, seed, F._case(seed), repr(e)[:300], flush=True)
print("done", lo, hi, "failures", bad)
