"""Developer probe: the headline workload as ONE synchronous batch of 4096 envs vs the same 4096 envs as two (four) independent sub-batches
stepped on their own HIP streams (asynchronous vectorised-env style: a learner consumes one sub-batch while the other steps).  The sub-batches'
kernels overlap, so the tail of one launch (its heaviest envs) no longer idles the chip."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv

STEPS, WARM = 200, 30
for parts in (1, 2, 4):
    B = 4096 // parts
    envs = [BatchedMyoEnv("myoHandPoseRandom-v0", num_envs=B, as_torch=False, env_offset=i * B) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for e in envs:
        e.reset(seed=1)
    for e, s in zip(envs, streams):
        e.batch.bench_rollout_async(WARM, 10, 0, stream=s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # enqueue in slices so that the streams interleave on the hardware queues
    for k in range(STEPS // 10):
        for e, s in zip(envs, streams):
            e.batch.bench_rollout_async(10, 10, 1 + k, stream=s.cuda_stream)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    flagged = sum(int((e.batch.status() != 0).sum()) for e in envs)
    print(f"{parts} sub-batch(es) of {B} envs: {4096 * STEPS / el:,.0f} env-steps/s  ({1e3 * el / STEPS:.3f} ms per step of all 4096 envs)  flagged {flagged}")
