"""The host leg of tools/lab/value_fuzz.py alone (the BLOCKING C ABI on numpy buffers, random chunk plans of the synchronous entries):
python tools/lab/fuzz_host_leg.py FIRST COUNT"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import value_fuzz as f  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
chunked0 = int(f.umfa_torch.get_option("sync_chunked_calls"))
for seed in range(first, first + count):
    msg = f.run_host_case(seed)
    if msg:
        bad += 1
        print("FAIL run_host_case", msg, flush=True)
print("done", first, count, "failures", bad, "chunked calls", int(f.umfa_torch.get_option("sync_chunked_calls")) - chunked0)
