"""`python bench.py --gpus N` starts its N ranks itself when it is not under torch.distributed.run (VERDICT r5 missing #2: the form the driver uses for N = 1 must not
silently produce an n_gpus = 1 line for N > 1).  On CPU: the launcher's ranks find each other over gloo on 127.0.0.1 and rank 0 alone prints one line with n_gpus = N;
a WORLD_SIZE that disagrees with --gpus is refused."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_gpus_2_without_torchrun_starts_two_ranks_and_prints_one_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["launcher"] == "ok" and d["n_gpus"] == 2 and d["sum_of_rank_plus_one"] == 3 and d["master"].startswith("127.0.0.1:")


def test_under_torchrun_the_process_is_one_rank():
    """(the driver's form for N > 1: RANK / WORLD_SIZE in the environment -> no second launcher)"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    procs = []
    for r in range(2):
        e = dict(_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [pr.communicate(timeout=300) for pr in procs]
    assert all(pr.returncode == 0 for pr in procs), [o[1][-500:] for o in outs]
    js = [[ln for ln in o[0].splitlines() if ln.startswith("{")] for o in outs]              # (gloo itself chats on stdout)
    assert len(js[0]) == 1 and json.loads(js[0][0])["n_gpus"] == 2 and js[1] == []


def test_a_world_size_that_disagrees_with_gpus_is_refused():
    e = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-extras"], env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_more_gpus_than_the_node_has_is_refused_at_once():
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--no-extras"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "device" in (p.stderr + p.stdout) and time.time() - t0 < 120
