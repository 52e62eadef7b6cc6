# quick A/B on the GPU box: headline bench twice + 32768 envs + parity tests; prints env-steps/s
for i in 1 2; do python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B4096 %.0f  kernel %.4f ms flagged %d' % (d['value'], d['roofline']['kernel_ms'], d['flagged_envs']))"; done
python bench.py --no-cpu-baseline --batch 32768 --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B32768 %.0f' % d['value'])"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_env.py -q -x 2>&1 | tail -2
