"""bench.py's own rank launcher (`python bench.py --gpus N` without torch.distributed.run): the sibling watchdog.
If a rank dies the launcher must stop the others -- only the processes it started -- and return non-zero within seconds,
instead of leaving rank 0 in a rendezvous until an outer time limit decides.  CPU test: the ranks are
stand-in scripts (the GPU version, with the real bench.py, is tests/test_gpu_multi.py::test_bench_dead_rank_is_noticed)."""
import importlib.util
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


RANK_SCRIPT = r"""
import os, sys, time
rank = int(os.environ["RANK"])
assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["LOCAL_RANK"] == str(rank)
mode = sys.argv[1]
if mode == "die" and rank == 1:
    time.sleep(0.5)
    sys.exit(7)
if mode == "ok":
    if rank == 0:
        print('{"metric": "x"}', flush=True)
    sys.exit(0)
time.sleep(600)            # a rank waiting for a peer that is gone
"""


def test_dead_rank_stops_the_run_within_seconds(capfd):
    bench = _bench()
    t = time.time()
    rc = bench.launch_ranks(3, cmd=[sys.executable, "-c", RANK_SCRIPT, "die"])
    dt = time.time() - t
    assert rc == 7 and dt < 20, (rc, dt)
    assert "rank 1 exited with code 7" in capfd.readouterr().err


def test_all_ranks_ok_forwards_rank0_stdout(capfd):
    bench = _bench()
    rc = bench.launch_ranks(3, cmd=[sys.executable, "-c", RANK_SCRIPT, "ok"])
    assert rc == 0
    assert capfd.readouterr().out.strip() == '{"metric": "x"}'


def test_help_text_formats():
    """argparse %-formats every help string: a bare per-cent sign in one of them made `bench.py --help` raise (round 5)."""
    import subprocess
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--gpus" in r.stdout, r.stderr[-500:]
