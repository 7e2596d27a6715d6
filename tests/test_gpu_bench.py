"""bench.py as the driver runs it, on the one GPU of the test box: every config's line, the strong-scaling mode, and the
N > 1 path with two ranks sharing the device (the launcher starts the ranks before anything touches the GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, nproc=1, timeout=900, own_launch=False):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if own_launch:   # the driver's command shape: plain python, bench.py starts its ranks itself
        env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), *args]
    elif nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args]
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def check_contract(o, n_gpus=1):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in o, k
    assert o["n_gpus"] == n_gpus and o["unit"] == "keys/s" and o["dtype"] == "u32" and o["vs_baseline"] is None
    assert "workload" in o["config"] and "model" not in o["config"]
    r = o["roofline"]
    assert r["peak"] == pytest.approx(39.3216) and "canonical_frac" in r
    if r["frac"] is not None:  # a PMC profile of exactly these kernel sources is committed
        assert 0.2 < r["frac"] < 1.3 and r["achieved"] == pytest.approx(r["frac"] * r["peak"])
    else:
        assert "pmc_refused" in r
    assert o["digest_ok"] is True and o["digests_checked_against_oracle"] >= 1


def test_default_config_two_steps_digest_checked_against_the_oracle():
    o = run_bench("--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    check_contract(o)
    assert o["config"]["baseline_config"] == 2 and o["config"]["layout"] == 1 and o["config"]["batch"] == 256
    assert o["matches"] == 130 and o["value"] > 1.0e9   # planted seeds in blocks 0 and 1
    assert o["digests_checked_against_oracle"] == 3       # the warm-up window + two 2^28 blocks of the full-space golden
    assert o["roofline_hbm"]["frac"] < 1.0
    # line hygiene: per-step spread, where engine creation went, registers from the code object
    ms = o["ms_per_step_rank0"]
    assert ms["min"] <= ms["median"] <= ms["max"] and ms["max"] < 1.5 * ms["min"]
    sp = o["engine_create_split_rank0"]
    assert sp["total_s"] == pytest.approx(sp["hip_runtime_s"] + sp["alloc_s"] + sp["table_wall_s"] + sp["other_s"]) and 0 < sp["table_build_s"] < 0.2
    if o["roofline"]["frac"] is not None:
        # the code object's own count: more than three waves per SIMD would allow (168), at most what two allow (256)
        assert 128 < o["roofline"]["vgpr"] <= 168 and o["roofline"]["waves_per_simd"] == 3   # the three-wave instantiation (compressed-only launch, automatic grid)
    # the one-sweep run that follows the timed region: the whole 2^32 space on the geometry the product picks for it
    s = o["strong"]
    assert s["covers_full_2p32"] is True and s["keys"] == 2**32 and s["matches"] == 1026
    assert (s["layout"], s["window_bits"], s["windows"]) == (1, 26, 10) and 18 < s["table_gb"] < 18.5     # ten signed windows
    assert s["digests_checked_against_oracle"] == 16 and s["wall_s"] > s["sweep_s"] > 1.5 and s["wall_s"] < 12
    import bench_workloads as wl
    assert s["digest"] == wl.load_golden("milksad_2p32.json")["digest"]


@pytest.mark.parametrize("cfg,steps,matches", [("3", 1, 1000), ("4", 2, None), ("5m", 1, 100), ("5e", 1, 100)])
def test_other_baseline_configs_have_a_bench_line(cfg, steps, matches):
    o = run_bench("--config", cfg, "--steps", str(steps), "--no-cpu-baseline")
    check_contract(o)
    assert o["config"]["baseline_config"] == int(cfg[0])
    if matches is not None:
        assert o["matches"] == matches
    if cfg.startswith("5"):
        assert o["inputs_per_s"] > 1e4 and o["value"] == pytest.approx(20 * o["inputs_per_s"], rel=1e-3)
        assert o["digests_checked_against_oracle"] == 2   # the warm-up inputs and the oracle's digest over all 100 000 inputs
    if cfg == "3":
        assert o["digests_checked_against_oracle"] == 2   # warm-up window + block 0 of the oracle's full-size golden
    if cfg == "4":
        assert o["digests_checked_against_oracle"] == 3   # warm-up window + blocks 0 and 1 of the oracle's 32-block golden


def test_strong_mode_reports_wall_time():
    o = run_bench("--strong", "--emulate-gpus", "8", "--layout", "0", "--window-bits", "26", "--no-cpu-baseline")
    check_contract(o)
    s = o["strong"]
    assert o["scaling"] == "strong" and o["steps"] == 2 and s["emulated_gpus"] == 8 and not s["covers_full_2p32"]
    assert s["wall_s"] > s["sweep_s"] > 0 and s["engine_create_s"] >= s["table_build_s"] > 0


def test_two_ranks_share_the_device():
    """The N > 1 path end to end (launcher, rank-major disjoint blocks, counter all-reduce, max-over-ranks time, digest
    gather) with both ranks on GPU 0.  RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the rehearsal's
    collectives run over gloo; on the driver's 8-GPU node the same code path runs with backend nccl, one rank per GPU."""
    o = run_bench("--share-device", "0", "--backend", "gloo", "--layout", "0", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-strong-after", nproc=2)
    check_contract(o, n_gpus=2)
    assert "strong" not in o
    import bench_workloads as wl
    planted = wl.c2_planted_seeds()
    blocks = (0, 1, 8, 9)                                                   # rank r starts at block r * 16/2
    assert o["matches"] == sum(1 for p in planted if (p >> 28) in blocks)
    assert o["digests_checked_against_oracle"] == 6
    g = wl.load_golden("milksad_2p32.json")
    assert o["digest"] == wl.hexd(wl.fold_digests([wl.unhexd(g["block_digests"][b]) for b in blocks]))
    assert "sweep_2p32_wall_s" not in o                                     # two steps are not a whole sweep
    o = run_bench("--share-device", "0", "--backend", "gloo", "--layout", "0", "--strong", "--emulate-gpus", "8", "--no-cpu-baseline", nproc=2)
    assert o["strong"]["blocks_per_rank"] == 2 and o["n_gpus"] == 2


def test_plain_gpus_2_starts_its_own_ranks():
    """The driver's command shape for N > 1 -- `python bench.py --gpus 2 ...`, no torch.distributed.run around it: bench.py starts
    the two ranks as a child process itself, rank 0's line comes through, blocks are disjoint and oracle-checked."""
    o = run_bench("--share-device", "0", "--backend", "gloo", "--layout", "0", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                  nproc=2, own_launch=True)
    check_contract(o, n_gpus=2)
    import bench_workloads as wl
    blocks = (0, 1, 8, 9)
    assert o["matches"] == sum(1 for p in wl.c2_planted_seeds() if (p >> 28) in blocks)
    g = wl.load_golden("milksad_2p32.json")
    assert o["digest"] == wl.hexd(wl.fold_digests([wl.unhexd(g["block_digests"][b]) for b in blocks]))
    assert o["digests_checked_against_oracle"] == 6 and o["engine_create_s"] > 0
    # the N-GPU line carries BOTH the sustained rate (value, 120 GB geometry -- here the fallback table, two ranks
    # share one card) and a full-sweep wall time the product would deliver: the two ranks share one 2^32 sweep on W = 24 tables
    s = o["strong"]
    assert s["covers_full_2p32"] is True and s["keys"] == 2**32 and s["blocks_per_rank"] == 8
    assert (s["layout"], s["window_bits"], s["windows"]) == (1, 24, 11) and 3.9 < s["table_gb"] < 4.1        # eleven signed windows
    assert s["matches"] == 1026 and s["digests_checked_against_oracle"] == 16 and s["digest"] == g["digest"]
    assert s["wall_s"] >= s["sweep_s"] > 0 and s["wall_s"] >= s["engine_create_s"]


def test_rccl_call_sequence_with_one_rank():
    """The collectives of the N > 1 path over RCCL itself (backend nccl: init with device_id, barrier, all_reduce SUM / MAX on
    device tensors, all_gather_object, destroy) on the one GPU of the box, world size 1 -- the two-rank rehearsal above has to
    use gloo because RCCL refuses two ranks on one device."""
    o = run_bench("--force-dist", "--layout", "0", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-strong-after")
    check_contract(o)
    assert o["n_gpus"] == 1 and o["matches"] == 65 and o["digests_checked_against_oracle"] == 2
