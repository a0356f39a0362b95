set -e
python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1 || { tail -20 gpurun_out/gputest.log; exit 1; }
tail -1 gpurun_out/gputest.log
for i in 1 2 3; do python bench.py --steps 1920 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long', d['value'], d['roofline']['frac'], d['kernel_ms'], d['parity'][:12])"; done
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('short', d['value'], d['roofline']['frac'])"; done
