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
blocks = [bench.block_of(rank, s, steps, block_log2, world=world) for s in range(steps)]
keys = matches = 0
dg = [0, 0, 0, 0]
m = vko.Matcher([vko.derive(vko.milksad_key(5)).p2pkh_compressed.decode(), vko.derive(vko.milksad_key((1 << 31) + 300)).p2pkh_compressed.decode()])
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
    # disjoint: rank r starts at block r * nblocks/world (here 2^24 blocks of 2^8 seeds), three blocks each
    half = 1 << 23
    assert sorted(blocks) == [(i * 256, i * 256 + 255) for i in (0, 1, 2, half, half + 1, half + 2)]
    assert out["tot"] == [6 * 256, 2] and out["tmax"] == 2.0


def test_block_partition_wraps_inside_the_seed_space():
    import bench
    seen = set()
    for rank in range(8):
        for s in range(2):
            lo, hi = bench.block_of(rank, s, 2, 28, world=8)
            assert hi - lo + 1 == 1 << 28 and hi < 1 << 32
            seen.add(lo)
    assert len(seen) == 16  # 8 ranks x 2 steps = the whole 2^32 space, no overlap
    assert bench.block_of(3, 14, 20, 28, world=4) == bench.block_of(0, (3 * 4 + 14) % 16, 20, 28, world=1)


def test_weak_run_contains_one_whole_sweep_for_every_driver_gpu_count():
    """The driver's line (--steps 20) at N = 1, 2, 4, 8: the first 16/N steps of the N ranks are disjoint and cover all 16
    blocks of the 2^32 space -- that prefix is what `sweep_2p32_wall_s` measures; a rank count that does not divide 16 falls
    back to rank-major blocks and reports no such time."""
    import bench
    for world in (1, 2, 4, 8, 16):
        n = 16 // world
        first = [bench.block_of(r, s, 20, 28, world=world)[0] >> 28 for r in range(world) for s in range(n)]
        assert sorted(first) == list(range(16))
        assert bench.weak_stride(20, 16, world) == n
    assert bench.weak_stride(20, 16, 3) == 20 and bench.weak_stride(20, 16, 32) == 20
    a = type("A", (), {"strong": False, "steps": 20})()
    for world, want in ((1, 16), (2, 8), (8, 2), (3, 0)):
        m = bench.Milksad(a, 0, world)
        m.nblocks = 16
        assert m.full_sweep_steps() == want
    a.steps = 4
    m = bench.Milksad(a, 0, 2)
    m.nblocks = 16
    assert m.full_sweep_steps() == 0   # the run is shorter than one sweep: nothing measured, nothing reported


def test_plain_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus N` with no launcher above it (the shape of the driver's N = 1 command) must start N ranks itself:
    the parent builds the torch.distributed.run command, runs it as a child and returns the child's status."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for extra in (["--strong"], ["--steps", "20", "--warmup", "2"], ["--config", "4"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", *extra, "--print-launch"],
                           capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        cmd = json.loads(r.stdout)
        i = cmd.index(os.path.join(ROOT, "bench.py"))
        assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd[:i] and "--nnodes=1" in cmd[:i]
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[i + 1:] == ["--gpus", "8", *extra]
    # under a launcher (WORLD_SIZE set) the same command does NOT start ranks again: it goes on to need a device
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--print-launch"], capture_output=True, text=True,
                       timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "torch.distributed.run" not in r.stdout


def test_self_launch_runs_the_ranks_as_a_child_and_returns_their_status():
    """Without a GPU the two ranks fail (no device); what is checked is that they were started -- both rank numbers appear in
    the launcher's failure report -- and that the parent hands the failure back instead of exiting 2 on an argument check."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_bench.py")
    assert r.returncode not in (0, 2), (r.returncode, r.stderr[-1500:])
    assert "rank      : 1" in r.stderr or "rank: 1" in r.stderr or "local_rank: 1" in r.stderr, r.stderr[-3000:]


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
