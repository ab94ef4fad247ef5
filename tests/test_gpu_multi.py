"""N > 1 on the GPU box.  The gpurun box has ONE MI355X, so several ranks / contexts share it:
that exercises the launcher, the sharding arithmetic, the halo copies and the stream ordering -- everything
except the xGMI links themselves.

* bench.py --gpus 2 starts its own two ranks (fresh child processes, the parent never touches the GPU) for both
  workloads; the N-rank output must equal the 1-rank output bit for bit (crc32 of every plane);
* the C ABI's multi-device entry points (srcnn_forward_y_striped[_dev], srcnn_forward_y_frames_multi) with 2-3
  contexts on cuda:0 against the single-context result, from Python and from a plain C++ host
  (tools/host_demo_multi.cpp)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_batch, synth_luma

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def run_bench(*args):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", "--steps", "3", "--warmup", "1",
                        *map(str, args)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_two_ranks_frames_equal_one_rank():
    """`python bench.py --gpus 2` as the README says it: rc 0, n_gpus 2, one frame per rank, no collective; the two
    output planes are frames 0 and 1 of the stream, exactly what one rank computes for --frames 2."""
    w, h = 1920, 1080
    one = run_bench("--gpus", 1, "--frames", 2, "--width", w, "--height", h)
    two = run_bench("--gpus", 2, "--shared-gpu", "--backend", "gloo", "--width", w, "--height", h)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and len(two["per_rank_ms_per_step"]) == 2
    assert two["distributed"]["halo_transport"].startswith("none")
    assert two["config"]["output_crc32"] == one["config"]["output_crc32"] and len(one["config"]["output_crc32"]) == 2
    assert two["value"] > 0 and two["roofline"]["frac"] > 0


@pytest.mark.parametrize("form", ["halo", "peer", "bands", "assemble"])
def test_bench_two_ranks_stripe_equal_one_rank(form):
    """configs[3] shape: one plane, two ranks, 6-row halo exchange; the stitched plane equals the 1-rank plane, whichever way
    a rank launches its step (one launch with the halo rows in buffers of their own -- the default --, one launch that reads
    the neighbour's rows through a HIP IPC mapping of its stripe with no per-step exchange at all, interior rows + edge bands,
    or one launch on an assembled copy)."""
    w, h = 1920, 1080
    one = run_bench("--gpus", 1, "--workload", "stripe", "--width", w, "--height", h)
    two = run_bench("--gpus", 2, "--shared-gpu", "--backend", "gloo", "--workload", "stripe", "--width", w, "--height", h,
                    *(["--no-overlap"] if form == "assemble" else ["--stripe-form", form]))
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["distributed"]["stripe_form"] == form and two["distributed"]["halo_overlap"] is (form != "assemble")
    assert two["config"]["output_crc32"] == one["config"]["output_crc32"] and len(one["config"]["output_crc32"]) == 1
    # ... and it is the plane the model of the kernel's arithmetic predicts
    import zlib
    m_out, _ = oracle.gpuorder_forward_y(synth_luma(w, h), S.load_weights())
    assert one["config"]["output_crc32"] == [zlib.crc32(m_out.tobytes())]


@pytest.fixture(scope="module")
def ctx_pool(weights_blob):
    ctxs = [S.Context(0) for _ in range(3)]
    for c in ctxs:
        c.set_weights_blob(weights_blob)
    yield ctxs
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("n_ctx,w,h", [(2, 260, 90), (3, 260, 90), (3, 131, 20), (2, 3840, 2160), (3, 1000, 37)])
def test_striped_over_contexts_equals_single_context(gpu_ctx, ctx_pool, n_ctx, w, h):
    """srcnn_forward_y_striped: own rows per context, ONE launch per stripe that reads the neighbours' 6 edge rows where they
    lie (stripes of 6 .. 17 rows included: 131x20 over 3, 1000x37 over 3) -- same bytes as one context."""
    y = synth_luma(w, h, frame=4)
    whole = gpu_ctx.forward_y(y)
    got = S.forward_y_striped(ctx_pool[:n_ctx], y)
    assert np.array_equal(got, whole)
    y2 = synth_luma(w, h, frame=5)
    assert np.array_equal(S.forward_y_striped(ctx_pool[:n_ctx], y2), gpu_ctx.forward_y(y2))


@pytest.mark.parametrize("n_ctx,w,h", [(2, 260, 90), (3, 131, 20), (3, 1920, 400)])
def test_striped_in_the_split_f16_mode_keeps_the_band_form(weights_blob, n_ctx, w, h):
    """The split-f16 kernels read one buffer only: in SRCNN_MODE_SPLIT16 the striped step still copies the halo rows on a
    second stream and launches interior rows + edge bands (thin stripes: one launch on an assembled copy) -- same bytes as one
    context in that mode, twice in a row (the band buffers of step 1 are reused by step 2, ordered by events)."""
    ctxs = [S.Context(0) for _ in range(n_ctx)]
    try:
        for c in ctxs:
            c.set_weights_blob(weights_blob)
            c.set_mode(S.MODE_SPLIT16)
        for frame in (4, 5):
            y = synth_luma(w, h, frame=frame)
            assert np.array_equal(S.forward_y_striped(ctxs, y), ctxs[0].forward_y(y))
    finally:
        for c in ctxs:
            c.close()


def test_bench_four_ranks_on_one_gpu():
    """The launcher, the gloo control plane (barrier, common start instant, gathers) and both workloads with FOUR ranks (they
    share the box's one GPU): output planes equal to what one rank computes."""
    w, h = 1280, 720
    one = run_bench("--gpus", 1, "--frames", 4, "--width", w, "--height", h)
    four = run_bench("--gpus", 4, "--shared-gpu", "--backend", "gloo", "--width", w, "--height", h)
    assert four["n_gpus"] == 4 and len(four["per_rank_ms_per_step"]) == 4
    assert four["config"]["output_crc32"] == one["config"]["output_crc32"]
    plane = run_bench("--gpus", 1, "--workload", "stripe", "--width", w, "--height", h)
    for form in ("halo", "peer"):
        striped = run_bench("--gpus", 4, "--shared-gpu", "--backend", "gloo", "--workload", "stripe", "--stripe-form", form,
                            "--width", w, "--height", h)
        assert striped["n_gpus"] == 4 and striped["config"]["output_crc32"] == plane["config"]["output_crc32"], form


def test_striped_dev_back_to_back(gpu_ctx, ctx_pool):
    """Device-resident striped steps queued back to back without host synchronisation in between."""
    import torch
    w, h, n_ctx = 700, 210, 3
    planes = [synth_luma(w, h, frame=f) for f in range(4)]
    rows = [S.stripe_rows(h, n_ctx, k) for k in range(n_ctx)]
    assert rows[0][0] == 0 and rows[-1][1] == h
    d_in = [[torch.from_numpy(p[a:b].copy()).cuda() for (a, b) in rows] for p in planes]
    d_out = [[torch.zeros_like(t) for t in step] for step in d_in]
    torch.cuda.synchronize()
    for step_in, step_out in zip(d_in, d_out):
        S.forward_y_striped_dev(ctx_pool[:n_ctx], [t.data_ptr() for t in step_in], w, [t.data_ptr() for t in step_out], w, w, h)
    for c in ctx_pool[:n_ctx]:
        c.synchronize()
    for p, step_out in zip(planes, d_out):
        got = np.concatenate([t.cpu().numpy() for t in step_out], axis=0)
        assert np.array_equal(got, gpu_ctx.forward_y(p))


def test_striped_step_reports_its_halo_transport(gpu_ctx, ctx_pool):
    """Contexts that share a device read each other's edge rows where they lie (transport 1; with peer access over xGMI it
    would be 2) -- no copy.  A link that refuses peer access (transport 3) is not an error, but the context says so
    (srcnn_halo_transport, srcnn_last_error): the copy path -- four halo buffer sets in turn on a second stream -- is forced
    here in a fresh process and must give the same bytes over many back-to-back steps."""
    y = synth_luma(700, 300, frame=4)
    S.forward_y_striped(ctx_pool[:3], y)
    assert [c.halo_transport() for c in ctx_pool[:3]] == [1, 1, 1]
    code = (
        "import numpy as np, torch, zlib, srcnn_cpp_amd as S\n"
        "from srcnn_cpp_amd.synth import synth_luma\n"
        "S.use_library(S.tuning_library_path())      # the knob below exists in the tuning build only\n"
        "blob = S.load_weights(); ctxs = [S.Context(0) for _ in range(3)]\n"
        "[c.set_weights_blob(blob) for c in ctxs]\n"
        "w, h = 700, 300\n"
        "crcs = []\n"
        "rows = [S.stripe_rows(h, 3, k) for k in range(3)]\n"
        "for f in range(11):\n"
        "    y = synth_luma(w, h, frame=f)\n"
        "    ins = [torch.from_numpy(np.ascontiguousarray(y[a:b])).cuda() for a, b in rows]\n"
        "    outs = [torch.zeros_like(t) for t in ins]\n"
        "    torch.cuda.synchronize()\n"
        "    for _ in range(3): S.forward_y_striped_dev(ctxs, [t.data_ptr() for t in ins], w, [t.data_ptr() for t in outs], w, w, h)\n"
        "    [c.synchronize() for c in ctxs]\n"
        "    crcs.append(zlib.crc32(np.concatenate([t.cpu().numpy() for t in outs]).tobytes()))\n"
        "print(crcs, [c.halo_transport() for c in ctxs], ctxs[1]._lib.srcnn_last_error(ctxs[1]._h).decode())\n")
    import os
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(ROOT),
                       env=dict(os.environ, SRCNN_DEBUG_HALO_STAGED="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    import zlib
    want = [zlib.crc32(gpu_ctx.forward_y(synth_luma(700, 300, frame=f)).tobytes()) for f in range(11)]
    assert r.stdout.strip().startswith(str(want)), r.stdout
    assert "[3, 3, 3]" in r.stdout


def test_frames_over_contexts_equal_single_context(gpu_ctx, ctx_pool):
    w, h, n = 300, 77, 7
    frames = synth_batch(w, h, n)
    ref = gpu_ctx.forward_y_frames(frames)
    for n_ctx in (2, 3):
        assert np.array_equal(S.forward_y_frames_multi(ctx_pool[:n_ctx], frames), ref)
    # more contexts than frames: the surplus contexts get an empty range
    assert np.array_equal(S.forward_y_frames_multi(ctx_pool, frames[:2]), ref[:2])


def test_multi_context_argument_errors(ctx_pool):
    y = synth_luma(64, 10)
    with pytest.raises(S.SrcnnError):
        S.forward_y_striped(ctx_pool[:2], y)                       # 5-row stripes < 6-row halo
    with pytest.raises(S.SrcnnError):
        S.forward_y_striped([ctx_pool[0], ctx_pool[0]], synth_luma(64, 40))     # the same context twice
    fresh = S.Context(0)
    try:
        with pytest.raises(S.SrcnnError):
            S.forward_y_striped([ctx_pool[0], fresh], synth_luma(64, 40))       # no model in the second context
    finally:
        fresh.close()


def test_cpp_host_with_several_contexts(tmp_path):
    """tools/host_demo_multi.cpp: a plain C++ host owning two contexts on cuda:0 runs one plane row-striped and a
    stream of frames through include/srcnn_amd.hpp; it compares both with its own single-context run bit for bit,
    and the last frame must be the plane the model predicts."""
    exe = tmp_path / "host_demo_multi"
    subprocess.run(["g++", "-std=c++17", "-pthread", f"-I{ROOT / 'include'}", str(ROOT / "tools" / "host_demo_multi.cpp"),
                    f"-L{ROOT / 'srcnn_cpp_amd'}", "-lsrcnn_amd", f"-Wl,-rpath,{ROOT / 'srcnn_cpp_amd'}",
                    "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    w, h, n = 520, 130, 5
    out_file = tmp_path / "out.u8"
    res = subprocess.run([str(exe), str(S._WEIGHTS_PATH), str(w), str(h), str(n), str(out_file), "0", "0"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr + res.stdout
    assert "on 2 contexts" in res.stdout and "lanes ok" in res.stdout
    got = np.fromfile(out_file, np.uint8).reshape(h, w)
    m_out, _ = oracle.gpuorder_forward_y(synth_luma(w, h, frame=n - 1), S.load_weights())
    assert np.array_equal(got, m_out)


def test_bench_dead_rank_is_noticed():
    """Rank 1 dies before the rendezvous; rank 0 would wait in init_process_group for the store timeout (30 min).
    The launcher's watchdog must stop it and return non-zero within seconds."""
    import time
    t = time.time()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", "--steps", "3", "--warmup", "1", "--gpus", "2",
                        "--shared-gpu", "--backend", "gloo", "--fault-rank", "1", "--width", "640", "--height", "360"],
                       capture_output=True, text=True, timeout=300)
    dt = time.time() - t
    assert r.returncode == 7 and dt < 30, (r.returncode, dt, r.stderr[-500:])
    assert "rank 1 exited with code 7" in r.stderr and not r.stdout.strip()


@pytest.mark.parametrize("workload", ["stripe", "frames"])
def test_bench_cxx_host_equals_python_ranks(workload):
    """`--host cxx`: ONE process, two contexts, the C ABI's several-GPUs entry points (hipMemcpyPeerAsync halo copies,
    persistent host threads) -- the second transport of the scaling curve.  Same planes as the one-rank run."""
    w, h = 1920, 1080
    extra = ["--workload", "stripe"] if workload == "stripe" else []
    one = run_bench("--gpus", 1, "--width", w, "--height", h, *(extra or ["--frames", 2]))
    two = run_bench("--gpus", 2, "--shared-gpu", "--host", "cxx", "--width", w, "--height", h, *extra)
    assert two["n_gpus"] == 2 and two["config"]["host"] == "cxx" and two["host_us_per_step"] > 0
    assert two["config"]["output_crc32"] == one["config"]["output_crc32"]
    assert two["scaling"] == ("strong" if workload == "stripe" else "weak")
    assert len(two["per_rank_ms_per_step"]) == 2 and two["value"] > 0


def test_bench_rccl_unavailable_fails_loudly_or_is_marked_degraded():
    """--backend nccl on a box where RCCL cannot build the group (two ranks on ONE GPU: RCCL refuses duplicate devices):
    the default is to give up with a non-zero exit code -- a number measured with the halo rows staged through host memory
    is not the configs[3] number -- and `--halo-fallback host` runs on, marking the line `"degraded": true`."""
    common = [sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", "--steps", "3", "--warmup", "1", "--gpus", "2",
              "--shared-gpu", "--backend", "nccl", "--workload", "stripe", "--width", "1920", "--height", "1080"]
    r = subprocess.run(common, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not any(l.startswith("{") for l in r.stdout.splitlines()), (r.returncode, r.stdout[-300:])
    assert "RCCL group unavailable" in r.stderr
    r = subprocess.run(common + ["--halo-fallback", "host"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])      # the JSON line is the LAST line, also with RCCL's chatter on stdout
    assert line["degraded"] is True and line["distributed"]["halo_transport"].startswith("host-staged")
    one = run_bench("--gpus", 1, "--workload", "stripe", "--width", 1920, "--height", 1080)
    assert line["config"]["output_crc32"] == one["config"]["output_crc32"]


def test_bench_line_reports_both_clock_regimes_and_e2e():
    """The default line carries the contract-as-written figure (`cold_start`: W warm-up + K steps before the pre-warm), the
    steady-state one, the PCIe-inclusive secondary metric of SURVEY 8d (`e2e`), and what the timed window is."""
    d = run_bench("--gpus", 1, "--width", 1920, "--height", 1080, "--steps", 10)
    assert d["cold_start"]["ms_per_step"] > 0 and d["cold_start"]["kernel_ms"] > 0
    assert d["e2e"]["value"] > 0 and d["e2e"]["output_equals_resident"] is True and d["e2e"]["value"] < d["value"]
    assert "closing_barrier_ms" in d["timing"]
    assert d["roofline"]["traffic"] is None or d["pmc_reference"]["traffic"] == d["roofline"]["traffic"]


def test_bench_under_torch_distributed_run():
    """The driver's own launch form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- every process is a rank (RANK / LOCAL_RANK / WORLD_SIZE in the
    environment), rank 0 prints the ONE line as the last line of its stdout.  Two ranks sharing cuda:0 here."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--shared-gpu", "--no-cpu-baseline", "--width", "1920", "--height", "1080"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and len(d["per_rank_ms_per_step"]) == 2
    assert d["timing"]["closing_barrier_ms"] >= 0 and d["cold_start"]["ms_per_step"] > 0
    one = run_bench("--gpus", 1, "--frames", 2, "--width", 1920, "--height", 1080)
    assert d["config"]["output_crc32"] == one["config"]["output_crc32"]
    # ... and the same line carries the row-striped configs[3] plane (RCCL refuses two ranks on one GPU: the halo form says so and
    # stages through the host; HSA_ENABLE_IPC_MODE_LEGACY is set by the worker itself, so the IPC form maps under this launcher too)
    st = d["stripe"]
    assert st["degraded"] is True and st["forms"]["halo"]["sha256_equals_golden"] is True
    assert st["forms"]["halo"]["halo_transport"].startswith("host-staged") and "rccl_error" in st["forms"]["halo"]
    assert st["forms"]["peer"].get("sha256_equals_golden") is True, st["forms"]["peer"]


def test_bench_eight_ranks_under_torch_distributed_run():
    """the driver's N = 8 command line AS IT IS -- torch.distributed.run, 8 ranks, default sizes: the frames
    workload on 3840x2160 planes (`value`) and, in the same line, the 7680x4320 plane of configs[3] in ITS partition, 8 stripes of
    540 rows, through both process-per-GPU transports.  The ranks share the box's one GPU, so RCCL cannot form the group (it
    refuses duplicate devices: the line must SAY so -- `rccl_world` null, `degraded`, the halo rows staged through the host) and
    no xGMI link is crossed; everything else of an 8-GPU run is exercised: the launcher's environment, the gloo control plane
    with 8 members, the common start instant, the gathers, the IPC mapping of both neighbours' stripes on interior ranks, the
    stitched plane's sha256 against the committed checksum."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--shared-gpu", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0 and len(d["per_rank_ms_per_step"]) == 8
    assert d["config"]["width"] == 3840 and len(d["config"]["output_crc32"]) == 8
    one = run_bench("--gpus", 1, "--frames", 8)
    assert d["config"]["output_crc32"] == one["config"]["output_crc32"]          # rank k computed frame k of the stream
    st = d["stripe"]
    gold = json.loads((ROOT / "tests" / "golden" / "config_checksums.json").read_text())["c3_7680x4320"]["gpuorder_sha256"][0]
    assert st["golden_sha256"] == gold
    for form in ("halo", "peer"):
        f = st["forms"][form]
        assert f["output_sha256"] == gold and f["sha256_equals_golden"] is True, (form, f)
        assert len(f["per_rank_ms"]) == 8 and f["ms_per_image"] > 0
    assert st["forms"]["halo"]["rccl_world"] is None and st["forms"]["halo"]["halo_transport"].startswith("host-staged")
    assert st["degraded"] is True and any("shared" in w for w in st["degraded_why"])


def test_bench_frames_line_carries_the_striped_plane():
    """ONE driver command per N must yield both halves of the metric.  `bench.py --gpus N` (frames, weak
    scaling: `value`) now also row-stripes the 7680x4320 plane of BASELINE configs[3] over its N ranks after the timed region
    and reports ms per image per transport in `stripe`: the stitched plane's sha256 is checked in-line against
    tests/golden/config_checksums.json, RCCL's own rank count and the transport are stated, anything else is `degraded`."""
    d = run_bench("--gpus", 2, "--shared-gpu", "--backend", "gloo", "--steps", 4)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["width"] == 3840
    one = run_bench("--gpus", 1, "--frames", 2, "--steps", 4)
    assert d["config"]["output_crc32"] == one["config"]["output_crc32"]          # the frames figure's planes are untouched
    st = d["stripe"]
    gold = json.loads((ROOT / "tests" / "golden" / "config_checksums.json").read_text())["c3_7680x4320"]["gpuorder_sha256"][0]
    assert st["golden_sha256"] == gold and st["one_gpu_reference"]["ms"] > 3
    for form in ("halo", "peer"):
        f = st["forms"][form]
        assert f["output_sha256"] == gold and f["sha256_equals_golden"] is True, (form, f)
        assert f["ms_per_image"] > 0 and len(f["per_rank_ms"]) == 2 and f["speedup_vs_one_gpu"] > 0
    assert st["forms"]["halo"]["halo_transport"] == "host-staged (gloo)" and st["forms"]["halo"]["rccl_world"] is None
    assert st["degraded"] is True and any("shared" in w for w in st["degraded_why"])       # two ranks on one GPU is a smoke configuration
    off = run_bench("--gpus", 2, "--shared-gpu", "--backend", "gloo", "--steps", 3, "--no-stripe-leg")
    assert "stripe" not in off


def test_a_hung_stripe_leg_does_not_take_the_frames_figure_with_it():
    """The stripe leg's transports have only ever run with the ranks sharing one GPU.  If one of them hangs on a real node the
    watchdog of the leg prints the line with the measured frames figure and an error in `stripe`, and every rank leaves with 0
    -- instead of the launcher's (or the driver's) time limit taking the whole line.  Rank 1 never enters the leg here; rank 0
    waits for it in the leg's first collective."""
    import time
    t = time.time()
    d = run_bench("--gpus", 2, "--shared-gpu", "--backend", "gloo", "--width", 1280, "--height", 720, "--stripe-timeout-s", 6,
                  "--hang-stripe-rank", 1)
    assert time.time() - t < 120
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["config"]["output_crc32"]) == 2
    assert "abandoned" in d["stripe"]["error"] and d["stripe"]["degraded"] is True


def test_peer_stripes_stream_planes_without_a_barrier(gpu_ctx, tmp_path):
    """`sharding.PeerStripeStep` streams NEW planes without a barrier per plane -- two stripe allocations per
    rank used in turn (both mapped by the neighbours once), `upload(k + 1)` while plane k computes, and only neighbour-to-
    neighbour handshakes ("uploaded" / "done with it") over gloo.  Three processes share cuda:0 and stream 8 different
    1920x1080 planes; every plane stitched from their rows equals the single-context result, and no rank entered a barrier
    between construction and close()."""
    import socket
    w, h, n_planes, world = 1920, 1080, 8, 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "helpers" / "peer_stream_rank.py"), str(w), str(h), str(n_planes),
                                       str(tmp_path / f"rank{r}.npy")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p_ in procs:
        try:
            o, e = p_.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()                                   # exactly the processes this test started
            raise
        outs.append((p_.returncode, o, e))
    assert all(rc == 0 for rc, _, _ in outs), [(rc, o[-300:], e[-1500:]) for rc, o, e in outs]
    assert all("barriers during the stream: 0" in o for _, o, _ in outs)
    parts = [np.load(tmp_path / f"rank{r}.npy") for r in range(world)]
    for g in range(n_planes):
        got = np.concatenate([p_[g] for p_ in parts], axis=0)
        assert np.array_equal(got, gpu_ctx.forward_y(synth_luma(w, h, frame=g))), g


@pytest.mark.parametrize("n_ctx,w,h,n_planes", [(3, 1920, 1080, 7), (2, 700, 210, 5), (3, 1280, 96, 6), (1, 640, 360, 3)])
def test_striped_frames_pipeline_equals_single_context(gpu_ctx, ctx_pool, n_ctx, w, h, n_planes):
    """srcnn_forward_y_striped_frames: a STREAM of different planes, each row-striped over the contexts, uploads / kernels /
    downloads of neighbouring planes overlapped and ordered across contexts by events only (the one-shot
    srcnn_forward_y_striped synchronises every context twice per plane).  Every plane equals the single-context result --
    thin stripes (24 rows), one context, an odd number of planes, the REFBYTES mode and a repeated call included."""
    planes = synth_batch(w, h, n_planes, first_frame=21)
    want = np.stack([gpu_ctx.forward_y(p) for p in planes])
    got = S.forward_y_striped_frames(ctx_pool[:n_ctx], planes)
    assert np.array_equal(got, want)
    assert np.array_equal(S.forward_y_striped_frames(ctx_pool[:n_ctx], planes[::-1].copy()), want[::-1])      # buffers reused
    if n_ctx == 2:
        for c in ctx_pool[:n_ctx]:
            c.set_mode(S.MODE_REFBYTES)
        try:
            ref = np.stack([oracle.forward_y(p, S.load_weights())[0] for p in planes])
            assert np.array_equal(S.forward_y_striped_frames(ctx_pool[:n_ctx], planes), ref)
        finally:
            for c in ctx_pool[:n_ctx]:
                c.set_mode(S.MODE_MFMA)
