"""One rank of tests/test_gpu_multi.py::test_peer_stripes_stream_planes_without_a_barrier (started by the test with RANK / WORLD_SIZE
/ MASTER_* set; every rank uses cuda:0).  Streams N different planes through sharding.PeerStripeStep -- upload(k + 1) right after
step(k) was queued, no dist.barrier between planes -- and writes this rank's output rows of every plane to an .npy file."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
import torch.distributed as dist

import srcnn_cpp_amd as S
from srcnn_cpp_amd import sharding
from srcnn_cpp_amd.synth import synth_luma

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
w, h, n_planes, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dist.init_process_group("gloo")
barriers = {"n": 0}
_barrier = dist.barrier


def counted(*a, **k):
    barriers["n"] += 1
    return _barrier(*a, **k)


dist.barrier = counted
torch.cuda.set_device(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = S.Context(0)
ctx.set_weights_blob(S.load_weights())
ctx.set_stream(stream.cuda_stream)
r0, r1 = sharding.stripe_rows(h, world, rank)
outs = [torch.zeros((r1 - r0, w), dtype=torch.uint8, device="cuda") for _ in range(n_planes)]
step = sharding.PeerStripeStep(ctx, synth_luma(w, h, frame=0, rows=(r0, r1)), outs[0], h, world, rank)
before = barriers["n"]
for g in range(n_planes):
    step.out = outs[g]
    step.step()                                           # plane g, queued
    if g + 1 < n_planes:
        step.upload(synth_luma(w, h, frame=g + 1, rows=(r0, r1)))      # plane g + 1 goes up while plane g computes
    if rank == 1 and g % 3 == 1:
        torch.cuda.synchronize()                          # ranks drift apart on purpose: the handshakes must hold them together
during = barriers["n"] - before
ctx.synchronize()
np.save(out_path, np.stack([o.cpu().numpy() for o in outs]))
step.close()
ctx.close()
print(f"rank {rank}: barriers during the stream: {during}", flush=True)
assert during == 0
dist.destroy_process_group()
