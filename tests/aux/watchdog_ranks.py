"""Driver of tests/test_bench_watchdog_cpu.py: one rank of a world-2 gloo job that hangs in its "probe" the first time (rank 1 never
enters the collective rank 0 waits in), is re-executed by bench._Watchdog with --exchange allreduce --fallback-from ..., rendezvous
again on the same MASTER_ADDR / MASTER_PORT and finishes a collective."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import bench

rank = int(os.environ["RANK"])
fallback = "--fallback-from" in sys.argv
dist.init_process_group("gloo")
t = torch.ones(1)
dist.all_reduce(t)
assert float(t) == 2.0
if not fallback:
    dog = bench._Watchdog(3.0, rank, "exchange probe 'rs_ag'", sys.argv, fallback="rs_ag")
    if rank == 0:
        dist.all_reduce(t)          # rank 1 never joins: this blocks until the watchdog re-executes the process
    else:
        time.sleep(3600)
    dog.cancel()
    sys.exit(9)                     # not reached
i = sys.argv.index("--exchange")
assert sys.argv[i + 1] == "allreduce" and sys.argv[sys.argv.index("--fallback-from") + 1] == "rs_ag"
assert "--exchange rs_ag" not in " ".join(sys.argv)
t = torch.full((1,), float(rank + 1))
dist.all_reduce(t)
print(f"rank {rank} fallback ok {float(t)}", flush=True)
dist.destroy_process_group()
