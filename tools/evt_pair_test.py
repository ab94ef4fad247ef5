"""HIP-event bracketing of the timed region: a pair around every step vs one pair around all steps (see bench.py)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_batch
W, H = 3840, 2160
ctx = S.Context(0); ctx.set_weights_blob(S.load_weights())
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
d_in = torch.from_numpy(synth_batch(W, H, 1)).cuda(); d_out = torch.zeros_like(d_in)
def step(): ctx.forward_y_dev(d_in.data_ptr(), W, H * W, d_out.data_ptr(), W, H * W, W, H, 1)
for _ in range(400): step()
torch.cuda.synchronize()
K = 50
for rep in range(3):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream); step(); b.record(stream)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / K * 1e3
    per = sum(a.elapsed_time(b) for a, b in ev) / K
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record(stream)
    for _ in range(K): step()
    b.record(stream)
    torch.cuda.synchronize(); wall2 = (time.perf_counter() - t0) / K * 1e3
    print(f"pair per step: events {per:.4f} ms, wall {wall:.4f} ms/step | one pair around {K} steps: events {a.elapsed_time(b) / K:.4f} ms, wall {wall2:.4f} ms/step")
