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


def test_strong_partition_is_one_disjoint_sweep():
    """bench.py --strong: N ranks x 16/N blocks, no wrap, together exactly the 2^32 space, for every N the driver uses."""
    import bench
    for world in (1, 2, 4, 8, 16):
        per = 16 // world
        blocks = [bench.block_of(r, s, per, 28, strong=True, world=world) for r in range(world) for s in range(per)]
        assert blocks == [(b << 28, ((b + 1) << 28) - 1) for b in range(16)]


def test_bench_fixtures_are_consistent_with_the_workload_definitions():
    """The committed oracle fixtures (tests/golden/bench_c*.json, milksad_planted.json) describe exactly the inputs
    bench_workloads.py generates: same planted seeds in the same order, same first synthetic input."""
    import bench_workloads as wl
    assert [p["seed"] for p in wl.load_golden("milksad_planted.json")["planted"]] == wl.c2_planted_seeds()
    c3 = wl.load_golden("bench_c3.json")
    assert [(p["seed"], p["sub_index"], p["addr_kind"]) for p in c3["planted"]] == wl.c3_planted()
    c4 = wl.load_golden("bench_c4.json")
    assert [(p["seed"], p["sub_index"]) for p in c4["planted"]] == wl.c4_planted() and len(c4["planted"]) == 1024
    c5 = wl.load_golden("bench_c5.json")
    assert c5["multibit"]["first_input"].encode() == wl.c5_mnemonics(1)[0] and c5["electrum"]["first_input"].encode() == wl.c5_electrum_seeds(1)[0]
    assert wl.c5_mnemonics(3, first=5) == wl.c5_mnemonics(8)[5:] and wl.c5_electrum_seeds(3, first=5) == wl.c5_electrum_seeds(8)[5:]
    assert [p["input"] for p in c5["multibit"]["planted"]] == wl.c5_planted_inputs() == [p["input"] for p in c5["electrum"]["planted"]]
    assert len(set(wl.c5_wordlist())) == 2048


def test_bench_fixture_spot_checks_against_the_oracle(vko):
    """A few entries of every bench fixture recomputed by the oracle (the generator is tests/golden/make_bench_fixtures.py)."""
    import bench_workloads as wl
    c3 = wl.load_golden("bench_c3.json")
    for p in c3["planted"][:5] + c3["planted"][-3:]:
        d = vko.derive(vko.transform_key(vko.spec(vko.T_SHA256), p["seed"], p["sub_index"]))
        assert (bytes(d.h160_u) if p["addr_kind"] == 1 else bytes(d.h160_c)).hex() == p["h160"]
    c4 = wl.load_golden("bench_c4.json")
    for p in c4["planted"][::200]:
        assert bytes(vko.derive(vko.transform_key(vko.spec(vko.T_LCG), p["seed"], p["sub_index"])).h160_c).hex() == p["h160"]
    pl = wl.load_golden("milksad_planted.json")["planted"]
    for p in pl[:3] + pl[-3:]:
        assert bytes(vko.derive(vko.milksad_key(p["seed"])).h160_c).hex() == p["h160_c"]
    c5 = wl.load_golden("bench_c5.json")
    p = c5["multibit"]["planted"][1]
    keys = vko.multibit_keys(vko.Wordlist(wl.c5_wordlist()), wl.c5_mnemonics(1, first=p["input"])[0])[2]
    assert bytes(vko.derive(keys[p["sub_index"]][1]).h160_c).hex() == p["h160"]
    st, _ = vko.sweep(vko.spec(vko.T_LCG), 0, 4095, flags=1 | vko.F_BATCHED, nthreads=2)
    assert st["keys"] == 4 * 4096


def test_golden_2p32_file_is_self_consistent_when_present():
    """tests/golden/milksad_2p32.json (hours of oracle CPU): windows fold into blocks, blocks into the full digest, and
    the committed 2^20-window fixture agrees with it where they overlap only through the fold rule."""
    import bench_workloads as wl
    g = wl.load_golden("milksad_2p32.json")
    if g is None:
        pytest.skip("full-space golden not generated yet")
    win = [wl.unhexd(d) for d in g["window_digests"]]
    assert len(win) == 256 and g["window_log2"] == 24
    blocks = [wl.fold_digests(win[16 * b:16 * b + 16]) for b in range(16)]
    assert [wl.hexd(b) for b in blocks] == g["block_digests"]
    assert wl.hexd(wl.fold_digests(blocks)) == g["digest"]
