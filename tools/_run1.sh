python -m pytest tests/test_gpu_parity.py tests/test_gpu_env.py tests/test_gpu_track.py -x -q 2>&1 | tail -3
b() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f kernel %.4f ms flagged %d' % (d['value'], d['roofline']['kernel_ms'], d['flagged_envs']))"; }
echo -n "hand4096 "; b --steps 300 --warmup 30
echo -n "hand4096 "; b --steps 300 --warmup 30
echo -n "hand32768 "; b --batch 32768 --steps 50 --warmup 10
echo -n "legs "; b --env myoLegWalk-v0 --steps 100 --warmup 20
echo -n "track "; b --env MyoHandAirplaneRandom-v0 --steps 100 --warmup 20
python tools/gpu_stamps.py > gpurun_out/r3_stamps3.txt 2>&1; tail -24 gpurun_out/r3_stamps3.txt
