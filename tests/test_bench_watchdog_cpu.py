"""bench.py's probe watchdog: a world-2 gloo job whose first "probe" hangs is re-executed rank by rank with the
plain all-reduce, rendezvous again on the launcher's address / port and completes - so that an exchange mode that hangs on the first
real 8-GPU node still yields a result line instead of a dead job."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_hung_probe_is_reexecuted_with_allreduce_world2():
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "aux", "watchdog_ranks.py"), "--exchange", "rs_ag", "--steps", "1"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, so[-500:], se[-1500:])
        assert f"rank {r} fallback ok 3.0" in so, (so, se[-800:])
        assert "watchdog: exchange probe 'rs_ag' did not finish in 3 s" in se
