timeout 2000 python -m pytest tests -m gpu -q 2>&1 | tail -4
python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['int8'])"
