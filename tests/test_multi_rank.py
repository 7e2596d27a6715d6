"""The N > 1 path of bench.py on CPU: world_size 2 over gloo.  The GPU work itself needs a device;
what is exercised here is everything around it -- the range partition (disjoint, covering), the
counter all-reduce and the max-over-ranks timing -- with the sweep replaced by the CPU oracle."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import bench
from oracle import vko
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
steps, block_log2 = 3, 8
blocks = [bench.block_of(rank, s, steps, block_log2) for s in range(steps)]
keys = matches = 0
dg = [0, 0, 0, 0]
m = vko.Matcher([vko.derive(vko.milksad_key(5)).p2pkh_compressed.decode(), vko.derive(vko.milksad_key(1300)).p2pkh_compressed.decode()])
for lo, hi in blocks:
    st, hits = vko.sweep(vko.spec(vko.T_MILKSAD), lo, hi, flags=1, matcher=m, nthreads=1)
    keys += st["keys"]; matches += st["matches"]
    for i in range(3): dg[i] ^= st["digest"][i]
tot, tmax = bench.reduce_counters(dist, torch.device("cpu"), [keys, matches], 1.0 + rank)
all_blocks = [None] * world
dist.all_gather_object(all_blocks, blocks)
if rank == 0:
    print(json.dumps({"tot": tot, "tmax": tmax, "blocks": all_blocks}))
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_partition_and_reduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), str(script), ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    blocks = [tuple(b) for rb in out["blocks"] for b in rb]
    # disjoint, contiguous cover of 6 blocks of 2^8 seeds
    assert sorted(blocks) == [(i * 256, i * 256 + 255) for i in range(6)]
    assert out["tot"] == [6 * 256, 2] and out["tmax"] == 2.0


def test_block_partition_wraps_inside_the_seed_space():
    import bench
    seen = set()
    for rank in range(8):
        for s in range(2):
            lo, hi = bench.block_of(rank, s, 2, 28)
            assert hi - lo + 1 == 1 << 28 and hi < 1 << 32
            seen.add(lo)
    assert len(seen) == 16  # 8 ranks x 2 steps = the whole 2^32 space, no overlap
    assert bench.block_of(3, 5, 16, 28) == bench.block_of(0, (3 * 16 + 5) % 16, 16, 28)
