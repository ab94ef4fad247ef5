"""Does the measured kernel time depend on how long the GPU has been busy?  Per-step HIP-event times of the fused 4K
launch after 0 / 5 / 50 / 500 warm-up steps (fresh context each time, 1 s idle in between)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, numpy as np
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_batch
W, H = 3840, 2160
ctx = S.Context(0); ctx.set_weights_blob(S.load_weights())
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
d_in = torch.from_numpy(synth_batch(W, H, 1)).cuda(); d_out = torch.zeros_like(d_in)
def step(): ctx.forward_y_dev(d_in.data_ptr(), W, H * W, d_out.data_ptr(), W, H * W, W, H, 1)
step(); torch.cuda.synchronize()
for warm in (0, 5, 50, 500, 5):
    time.sleep(1.0)
    for _ in range(warm): step()
    K = 20
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    for a, b in ev:
        a.record(stream); step(); b.record(stream)
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in ev]
    print(f"warm-up {warm:4d} steps: mean {np.mean(ts):.4f} ms; steps 1-5 {np.mean(ts[:5]):.4f}, 6-10 {np.mean(ts[5:10]):.4f}, 16-20 {np.mean(ts[15:]):.4f}")
