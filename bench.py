#!/usr/bin/env python3
"""bench.py -- keys/sec of the seed -> key -> k*G -> hash160 -> probe path on MI355X, one JSON line on rank 0.

Default workload = BASELINE.json configs[1]: Milksad (MT19937) seed sweep, compressed P2PKH, 1024 (+2) planted targets
in the on-device probe table; one *step* = one 2^28-seed block of the 2^32 seed space per GPU (16 steps on one GPU =
the whole sweep).  With N GPUs, rank r sweeps blocks (r*K + step) mod 16: the seed space is range-partitioned, no
data-path collective; RCCL is used only to all-reduce counters after the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5m|5e] [--strong]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
A plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) starts those N ranks itself: the parent, before it
imports torch or touches the GPU, runs the torch.distributed.run line above as a child process and returns its status.

Other configs under the same contract (inputs in bench_workloads.py):
  --config 3   scan --transform=sha256 vs 1 000 000 targets (both hash160s), step = 2^25 inputs (3 keys each)
  --config 4   all LCG variants, step = 2^26 seeds (4 keys each) of the 2^31 seed space, range-split over ranks
  --config 5m  100 000 synthetic MultiBit-HD mnemonics -> 20 keys each -> probe (vk_scan_strings), lines split over ranks
  --config 5e  the same for Electrum v1 seeds
--strong (config 2): the N ranks share ONE 2^32 sweep (16/N disjoint blocks each, no wrap) and the line also reports the
end-to-end wall time from before the engine (table build) exists to after the counters are reduced.
The default config-2 line carries that measurement too, as `strong`: AFTER the timed region (never inside it, never part of
`value`; the 120 GB engine stays alive beside it) the same ranks share one whole 2^32 sweep on the geometry the product picks for a
single sweep (ten signed windows = 18 GB on one GPU, eleven = 4 GB from two), timed from before vk_engine_create to after the counter
all-reduce -- `strong.wall_s`, `covers_full_2p32: true` (--no-strong-after skips it).

Every expected value comes from the CPU oracle through committed fixtures (tests/golden/): planted hash160s, the
digest of every 2^28 block of the Milksad space, the digest of each config's warm-up window.  bench.py itself touches
oracle/ only in the cpu_baseline leg.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Integer-VALU peak: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = 39.3 T lane-ops/s.  This is the
# 16-lanes-per-clock class that carries this path's arithmetic (v_mad_u64_u32, v_addc_co_u32, v_add3, v_alignbit:
# 30-38 T measured, profiles/archive/r01_ubench_valu_rates.txt); plain VOP2 adds / logic issue ~1.5x faster (52-58 T) and make
# up ~15 % of the mix, so a fraction slightly above the naive bound is possible in principle.
PEAK_INT32 = 256 * 4 * 16 * 2.4e9
HBM_PEAK = 8.0e12


def weak_stride(steps, nblocks, world):
    """Blocks between the starting points of neighbouring ranks in the weak-scaling run.  When the rank count divides the block
    count, rank r starts at r*nblocks/world: the first nblocks/world steps of all ranks together are then exactly ONE sweep of
    the whole space (disjoint blocks), which is what `sweep_2p32_wall_s` times; later steps wrap.  Otherwise rank-major."""
    return nblocks // world if (world <= nblocks and nblocks % world == 0) else steps


def block_of(rank, step, steps, block_log2=28, strong=False, world=1):
    """Seed block (inclusive lo, hi) of the 2^32 Milksad space that `rank` sweeps at `step`.  Weak scaling (driver default):
    rank r walks blocks r*16/world, r*16/world + 1, ... wrapping inside the seed space (per-GPU work fixed at `steps` blocks).
    --strong: the world shares ONE sweep, rank r owns the contiguous run of 16/world blocks starting at r*16/world and stops
    there (disjoint, no wrap, together the whole space)."""
    nblocks = 1 << (32 - block_log2)
    b = (rank * (nblocks // world) + step) if strong else (rank * weak_stride(steps, nblocks, world) + step) % nblocks
    return b << block_log2, ((b + 1) << block_log2) - 1


def launcher_cmd(gpus, argv, port):
    """The command a plain `bench.py --gpus N` runs as a child: one rank per GPU under torch.distributed.run, same arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher above it: start the N ranks as a CHILD process (never an exec, and
    before this process imports torch or makes any GPU call), let rank 0's JSON line through and hand back the child's status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = launcher_cmd(args.gpus, [a for a in argv if a != "--print-launch"], port)
    if args.print_launch:
        print(json.dumps(cmd))
        return 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("OMP_NUM_THREADS", "4")
    # rank 0's JSON line goes to stdout; anything else a rank or a library writes there (gloo announces its connections on
    # stdout) is passed on to stderr, so that stdout carries exactly the one line the contract promises
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in p.stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        (sys.stdout if line.startswith("{") else sys.stderr).flush()
    return p.wait()


def csrc_hash():
    """SHA-256 over the kernel sources with comments and white space removed: PMC-derived numbers are only valid for the exact CODE
    they were taken from (a reworded comment does not re-key the profiles, any token change does)."""
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vuke_amd", "csrc")
    strip = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"', re.S)
    for f in sorted(os.listdir(d)):
        if f.endswith((".cuh", ".hip", ".h")):
            txt = open(os.path.join(d, f), encoding="utf-8", errors="replace").read()
            txt = strip.sub(lambda m: m.group(0) if m.group(0).startswith('"') else " ", txt)   # string literals are code
            h.update(f.encode())
            h.update(" ".join(txt.split()).encode())
    return h.hexdigest()[:16]


def load_pmc(tag, kernel=None):
    """profiles/pmc_<tag>.json (tools/pmc_summary.py) if it was taken from the kernel sources in this tree, else a reason."""
    path = os.path.join(ROOT, "profiles", f"pmc_{tag}.json")
    if not os.path.exists(path):
        return None, f"no profiles/pmc_{tag}.json"
    pj = json.load(open(path))
    if pj.get("csrc_hash") != csrc_hash():
        return None, f"profiles/pmc_{tag}.json was taken from other kernel sources ({pj.get('csrc_hash')} != {csrc_hash()}): re-run tools/profile.sh"
    if kernel and kernel not in pj.get("kernels", []):
        return None, f"profiles/pmc_{tag}.json describes {pj.get('kernels', ['?'])[0]}, this run launches {kernel} (another --blocks-per-cu): no roofline for it"
    floor, meas = pj.get("cycles_per_valu_inst_floor"), pj.get("cycles_per_valu_inst_per_simd")
    if floor and meas and floor > meas:
        return None, (f"profiles/pmc_{tag}.json prices the instruction mix at {floor:.3f} cycles per instruction, ABOVE the {meas:.3f} it measured: "
                      "the floor model is wrong for this kernel (tools/pmc_summary.py, profiles/valu_rates.json)")
    return pj, None


def kernel_meta(kernel):
    """Code-object metadata of `kernel` from profiles/kernel_meta.json (tools/kernel_meta.py) if it was taken from the kernel
    sources in this tree, else None."""
    path = os.path.join(ROOT, "profiles", "kernel_meta.json")
    if not os.path.exists(path):
        return None
    km = json.load(open(path))
    if km.get("csrc_hash") != csrc_hash():
        return None
    return next((r for r in km["kernels"] if kernel in r["kernel"]), None)


def sweep_kernel_name(kind, flags, blocks_per_cu):
    """The instantiation of vk_sweep_kernel<KIND, WAVES, UNCOMP, WIDE> that runs a launch: a mirror of the engine's choice
    (vuke_amd/csrc/vk_sweep_tu.hip: vk_sweep_design_form; vk_engine.hip: run_range), held to it on the GPU box through the
    vk_debug_last_launch test hook (tests/test_gpu_bench.py).  The metadata and the PMC summary the roofline block quotes are
    looked up under this name, so a run with another --blocks-per-cu prices the kernel it actually ran."""
    three_wave_kinds = (1, 2, 3, 4, 100)   # milksad, mt64, lcg, xorshift, raw keys
    if blocks_per_cu == 0 or blocks_per_cu >= 3:
        if flags & 2:
            return f"vk_sweep_kernel<{kind}u, 3, true, true>"
        if kind in three_wave_kinds:
            return f"vk_sweep_kernel<{kind}u, 3, false, false>"
    return f"vk_sweep_kernel<{kind}u, 2, true, false>"


# ------------------------------------------------------------------------------------------------ workloads
class Workload:
    """One BASELINE config: targets, warm-up (checked against the oracle's fixture), timed steps."""
    tag = "?"
    kernel = "?"
    unit_name = "keys"

    def __init__(self, args, rank, world):
        self.args, self.rank, self.world = args, rank, world
        self.digest_checked = self.digest_bad = 0
        if getattr(self, "sweep_kind", None) is not None:
            self.kernel = sweep_kernel_name(self.sweep_kind, self.sweep_flags, getattr(args, "blocks_per_cu", 0))

    def check_digest(self, got, want):
        self.digest_checked += 1
        if tuple(got) != tuple(want):
            self.digest_bad += 1

    def step_total(self):
        return self.args.steps


class Milksad(Workload):
    tag, sweep_kind, sweep_flags = "c2", 1, 1   # the kernel name follows from kind, flags and --blocks-per-cu (sweep_kernel_name)
    # canonical algorithmic work per Milksad key (32-bit integer ops of a 4-bit-window reference algorithm)
    canonical_ops_per_key = 1736 + 213000 + 2168 + 1455

    def describe(self, eng):
        return ("Milksad (MT19937) seed sweep, P2PKH compressed, 1026 planted targets; step = one 2^28-seed block per GPU "
                "(16 blocks = full 2^32 sweep)" + ("; --strong: the ranks share one 2^32 sweep" if self.args.strong else ""))

    def setup(self, eng, ffi, wl):
        self.ffi, self.wl = ffi, wl
        self.spec = ffi.spec(ffi.T_MILKSAD)
        pl = wl.load_golden("milksad_planted.json")["planted"]
        self.planted = pl
        self.pseeds = [p["seed"] for p in pl]
        eng.targets_set([bytes.fromhex(p["h160_c"]) for p in pl], [ffi.TARGET_P2PKH] * len(pl))
        g = wl.load_golden("milksad_2p32.json")
        self.block_digest = [wl.unhexd(d) for d in g["block_digests"]] if g else None
        self.win = {w["lo"]: w for w in (wl.load_golden("digest_windows.json") or [])}
        self.nblocks = 1 << (32 - wl.C2_BLOCK_LOG2)
        if self.args.strong:
            assert self.nblocks % self.world == 0, "--strong needs a GPU count that divides 16"
            self.per_rank = self.nblocks // self.world

    def step_total(self):
        return self.per_rank if self.args.strong else self.args.steps

    def full_sweep_steps(self):
        """Steps after which all ranks together have swept every block of the 2^32 space exactly once (0 = this run never
        does): 16/world when the rank count divides 16 (block_of: rank r starts at block r*16/world) and the run is that long."""
        if self.nblocks % self.world or self.world > self.nblocks:
            return 0
        n = self.nblocks // self.world
        return n if self.step_total() >= n else 0

    def block_of(self, step):
        lo, _ = block_of(self.rank, step, self.args.steps, self.wl.C2_BLOCK_LOG2, self.args.strong, self.world)
        return lo >> self.wl.C2_BLOCK_LOG2

    def warmup(self, eng):
        # the first 2^20 seeds: the oracle's digest for that window is a committed fixture (tests/golden/digest_windows.json)
        st, _ = eng.sweep(self.spec, 0, (1 << 20) - 1, flags=self.ffi.F_COMPRESSED, hit_cap=2048)
        w = self.win.get(0)
        if w:
            self.check_digest(st["digest"], self.wl.unhexd(w["digest"]))
        lo = self.block_of(0) << self.wl.C2_BLOCK_LOG2
        eng.sweep(self.spec, lo, lo + (1 << 24) - 1, flags=self.ffi.F_COMPRESSED, hit_cap=2048)  # clocks and caches up
        return st

    def step(self, eng, s):
        b = self.block_of(s)
        lo, hi = b << self.wl.C2_BLOCK_LOG2, ((b + 1) << self.wl.C2_BLOCK_LOG2) - 1
        st, hits = eng.sweep(self.spec, lo, hi, flags=self.ffi.F_COMPRESSED, hit_cap=2048)
        if self.block_digest:
            self.check_digest(st["digest"], self.block_digest[b])
        st["expected"] = sum(1 for p in self.pseeds if lo <= p <= hi)
        want = [(p["seed"], p["privkey"], p["h160_c"]) for p in self.planted if lo <= p["seed"] <= hi]
        if [(h["seed"], h["privkey"].hex(), h["h160"].hex()) for h in hits] != want:
            st["matches"] = -1  # wrong hit records
        st["units"] = st["keys"]
        return st

    def algo_bytes_per_key(self, info):
        return info["windows"] * 64 + 128 + 160 + 32  # table entries + workspace write + workspace read + probe line


class Sha256Scan(Workload):
    tag, sweep_kind, sweep_flags = "c3", 0, 3
    canonical_ops_per_key = 2168 + 213000 + 3 * 2168 + 2 * 1455

    def describe(self, eng):
        return "scan --transform=sha256 vs 1 000 000 targets (1000 planted + 999 000 decoys), both hash160s, 3 keys/input; step = 2^25 inputs per GPU"

    def setup(self, eng, ffi, wl):
        self.ffi, self.wl = ffi, wl
        self.spec = ffi.spec(ffi.T_SHA256)
        self.fx = wl.load_golden("bench_c3.json")
        h, k = [], []
        for p in self.fx["planted"]:
            h.append(bytes.fromhex(p["h160"]))
            k.append(ffi.TARGET_P2WPKH if p["addr_kind"] == 2 else ffi.TARGET_P2PKH)
        for d, kind in wl.c3_decoys():
            h.append(d)
            k.append(kind)
        eng.targets_set(h, k)
        self.pseeds = [p["seed"] for p in self.fx["planted"]]
        full = wl.load_golden("bench_c3_full.json")   # the oracle over the config's whole input range, block by block
        self.block_digest = {b["lo"]: wl.unhexd(b["digest"]) for b in full["blocks"]} if full else {}

    def warmup(self, eng):
        w = self.fx["warmup"]
        st, _ = eng.sweep(self.spec, w["lo"], w["hi"], flags=3, hit_cap=2048)
        self.check_digest(st["digest"], self.wl.unhexd(w["digest"]))
        return st

    def step(self, eng, s):
        n = self.wl.C3_STEP_INPUTS
        b = self.rank * self.args.steps + s
        lo, hi = b * n, (b + 1) * n - 1
        st, hits = eng.sweep(self.spec, lo, hi, flags=3, hit_cap=2048)
        if lo in self.block_digest:
            self.check_digest(st["digest"], self.block_digest[lo])
        st["expected"] = sum(1 for p in self.pseeds if lo <= p <= hi)
        want = sorted((p["seed"], p["sub_index"], p["addr_kind"], p["privkey"], p["h160"]) for p in self.fx["planted"] if lo <= p["seed"] <= hi)
        if [(h["seed"], h["sub_index"], h["addr_kind"], h["privkey"].hex(), h["h160"].hex()) for h in hits] != want:
            st["matches"] = -1  # wrong hit records (seed, slot, address kind, key, hash160): fails the parity check
        st["units"] = st["keys"]
        return st

    def algo_bytes_per_key(self, info):
        return info["windows"] * 64 + 128 + 160 + 2 * 32


class LcgAll(Workload):
    tag, sweep_kind, sweep_flags = "c4", 3, 1
    canonical_ops_per_key = 40 + 213000 + 2168 + 1455

    def describe(self, eng):
        return "all LCG variants (glibc, minstd, msvc, borland; BE) x 2^31 seeds, compressed; step = one 2^26-seed block (2^28 keys) per GPU, 32 blocks range-split over the ranks"

    def setup(self, eng, ffi, wl):
        self.ffi, self.wl = ffi, wl
        self.spec = ffi.spec(ffi.T_LCG)
        self.fx = wl.load_golden("bench_c4.json")
        pl = self.fx["planted"]
        eng.targets_set([bytes.fromhex(p["h160"]) for p in pl], [ffi.TARGET_P2PKH] * len(pl))
        self.pseeds = [p["seed"] for p in pl]
        self.nblocks = wl.C4_SEEDS // wl.C4_STEP_SEEDS
        full = wl.load_golden("bench_c4_full.json")   # the oracle over all four variants x the whole 2^31 seed space
        self.block_digest = {b["lo"]: (b["keys"], wl.unhexd(b["digest"])) for b in full["blocks"]} if full else {}

    def warmup(self, eng):
        w = self.fx["warmup"]
        st, _ = eng.sweep(self.spec, w["lo"], w["hi"], flags=1, hit_cap=2048)
        self.check_digest(st["digest"], self.wl.unhexd(w["digest"]))
        return st

    def step(self, eng, s):
        n = self.wl.C4_STEP_SEEDS
        b = (self.rank * self.args.steps + s) % self.nblocks
        lo, hi = b * n, (b + 1) * n - 1
        st, hits = eng.sweep(self.spec, lo, hi, flags=1, hit_cap=2048)
        want = sorted((p["seed"], p["sub_index"], p["addr_kind"], p["privkey"], p["h160"]) for p in self.fx["planted"] if lo <= p["seed"] <= hi)
        if [(h["seed"], h["sub_index"], h["addr_kind"], h["privkey"].hex(), h["h160"].hex()) for h in hits] != want:
            st["matches"] = -1  # wrong hit records
        if lo in self.block_digest:
            keys, dg = self.block_digest[lo]
            self.check_digest((st["keys"],) + tuple(st["digest"]), (keys,) + dg)
        st["expected"] = sum(1 for p in self.pseeds if lo <= p <= hi)
        st["units"] = st["keys"]
        return st

    def algo_bytes_per_key(self, info):
        return info["windows"] * 64 + 128 + 160 + 32


class Chains(Workload):
    """config 5: string inputs -> KDF chain -> 20 keys -> derive -> probe, fused on the device (vk_scan_strings)."""
    unit_name = "inputs"

    def __init__(self, args, rank, world, which):
        super().__init__(args, rank, world)
        self.which = which
        self.tag = "c5m" if which == "multibit" else "c5e"
        self.kernel = "vk_ring_kernel<vk_pbkdf2_body>" if which == "multibit" else "vk_ring_kernel<vk_electrum_body>"
        # SHA-512 compression 5520 ops; MultiBit = 2 x 2048 x 2 compressions + 40 k*G + 20 derives;
        # Electrum = 200 000 SHA-256 compressions + 21 k*G (+ 20 derives)
        self.canonical_ops_per_input = (2 * 2048 * 2 * 5520 + 40 * 213000 + 20 * 220000) if which == "multibit" else (200000 * 2168 + 21 * 213000 + 20 * 220000)

    def describe(self, eng):
        return (f"{self.wl.C5_INPUTS} synthetic {'12-word MultiBit-HD mnemonics' if self.which == 'multibit' else '32-hex-digit Electrum v1 seeds'} "
                f"-> 20 keys each -> both hash160s -> probe (100 planted: key 7 of every 1000th input); step = one pass over the rank's slice of the inputs")

    def setup(self, eng, ffi, wl):
        self.ffi, self.wl = ffi, wl
        fx = wl.load_golden("bench_c5.json")[self.which]
        self.fx = fx
        if self.which == "multibit":
            eng.bip39_wordlist_set(wl.c5_wordlist())
            self.spec = ffi.spec(ffi.T_MULTIBIT, 0, 0, 20)
            lines = wl.c5_mnemonics()
            nwarm = wl.C5_WARMUP_MULTIBIT
        else:
            self.spec = ffi.spec(ffi.T_ELECTRUM, 0, 0, 20)
            lines = wl.c5_electrum_seeds()
            nwarm = wl.C5_WARMUP_ELECTRUM
        assert lines[0].decode() == fx["first_input"]
        pl = fx["planted"]
        eng.targets_set([bytes.fromhex(p["h160"]) for p in pl], [ffi.TARGET_P2PKH] * len(pl))
        # line partition: rank r takes the r-th contiguous slice of the same 100 000 inputs
        n = len(lines)
        a, b = n * self.rank // self.world, n * (self.rank + 1) // self.world
        self.first = a
        self.packed = ffi.pack_strings(lines[a:b])
        self.warm_packed = ffi.pack_strings(lines[:nwarm])
        self.expected = sum(1 for p in pl if a <= p["input"] < b)
        self.planted_here = [(p["input"] - a, p["sub_index"]) for p in pl if a <= p["input"] < b]
        full = wl.load_golden("bench_c5_full.json")   # the oracle over all 100 000 inputs (one rank sweeps them all only at N = 1)
        self.full = full[self.which] if (full and self.world == 1) else None

    def warmup(self, eng):
        st, _ = eng.scan_strings(self.spec, None, flags=3, hit_cap=2048, packed=self.warm_packed)
        w = self.fx["warmup"]
        self.check_digest(st["digest"], self.wl.unhexd(w["digest"]))
        return st

    def step(self, eng, s):
        st, hits = eng.scan_strings(self.spec, None, flags=3, hit_cap=2048, packed=self.packed)
        st["expected"] = self.expected
        if self.full:
            self.check_digest((st["keys"],) + tuple(st["digest"]), (self.full["keys"],) + self.wl.unhexd(self.full["digest"]))
        if [(h["index"], h["sub_index"]) for h in hits] != self.planted_here:
            st["matches"] = -1  # wrong hit set: fails the parity check below
        st["units"] = st["inputs"]
        return st

    def algo_bytes_per_key(self, info):
        return None


def make_workload(args, rank, world):
    c = args.config
    if c == "2":
        return Milksad(args, rank, world)
    if c == "3":
        return Sha256Scan(args, rank, world)
    if c == "4":
        return LcgAll(args, rank, world)
    if c in ("5m", "5e"):
        return Chains(args, rank, world, "multibit" if c == "5m" else "electrum")
    raise SystemExit(f"bench.py: unknown --config {c}")


# ------------------------------------------------------------------------------------------------ main
def reduce_counters(dist, dev, counters, elapsed):
    """Sum the integer counters and take the max over ranks of the elapsed time(s) (one tiny collective each).
    `elapsed` is one float (returned as a float) or a list of floats (returned as a list)."""
    import torch
    many = isinstance(elapsed, (list, tuple))
    tot = torch.tensor(list(counters), dtype=torch.int64, device=dev)
    tmax = torch.tensor(list(elapsed) if many else [elapsed], dtype=torch.float64, device=dev)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return [int(x) for x in tot.tolist()], ([float(x) for x in tmax.tolist()] if many else float(tmax.item()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="2", choices=["2", "3", "4", "5m", "5e"])
    ap.add_argument("--strong", action="store_true", help="config 2: N ranks share one 2^32 sweep; also report end-to-end wall time")
    ap.add_argument("--emulate-gpus", type=int, default=0, help="--strong on one GPU: sweep only this rank-0 share of an M-GPU run (16/M blocks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong-after", action="store_true", help="config 2: skip the one-whole-sweep measurement that follows the timed region")
    # rehearsal knobs (not used by the driver): run the N-rank path on a box with fewer GPUs than ranks
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a rehearsal)")
    ap.add_argument("--share-device", type=int, default=-1, help="put every rank on this one GPU (rehearsal only)")
    ap.add_argument("--print-launch", action="store_true", help="with --gpus N > 1 and no launcher: print the child command as JSON and exit")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run every collective even with one rank "
                                                               "(exercises the RCCL call sequence of the N > 1 path on a one-GPU box)")
    # table geometry of the flagship run: 9 signed-digit windows (120 GB of the 288 GB HBM, 8 point additions per key),
    # 256 keys per lane per field inversion, 2 resident workgroups per CU.  --layout 0 --window-bits 24 is the library
    # default (11.8 GB, 10 additions).
    ap.add_argument("--layout", type=int, default=None)
    ap.add_argument("--window-bits", type=int, default=None)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--blocks-per-cu", type=int, default=0, help="workgroups per CU; 0 = per launch the form the engine picks (three waves per SIMD: three 256-lane workgroups per CU, or one of 768 lanes where uncompressed keys are hashed); 2 = the two-wave form")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.steps is None:
        args.steps = {"2": 16, "3": 3, "4": 8, "5m": 10, "5e": 3}[args.config]
    if args.strong and args.config != "2":
        raise SystemExit("bench.py: --strong is defined for --config 2")
    world_hint = args.emulate_gpus or int(os.environ.get("WORLD_SIZE", "1"))
    if args.layout is None and args.window_bits is None:
        if args.strong:
            # ONE 2^32 sweep end to end: allocating and building the 120 GB table costs 0.4-5 s (hipMalloc of 120 GB varies
            # that much between boxes), more than it saves; by measured wall time (profiles/archive/r02_strong_geometry.txt) the
            # ten-window table wins on one GPU and the eleven-window table from two GPUs up
            args.layout, args.window_bits = 1, (26 if world_hint == 1 else 24)   # signed windows: 18 GB / 4 GB (round 4; unsigned 43 / 11.8 GB before)
        else:
            args.layout, args.window_bits = 1, 0   # sustained throughput: 9 signed windows, 120 GB
    args.layout = args.layout or 0
    args.window_bits = args.window_bits or 0

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_device >= 0:
        local_rank = args.share_device
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    if args.gpus != world and not args.force_dist:
        print(f"bench.py: --gpus {args.gpus} under a launcher with WORLD_SIZE={world}: the launcher's rank count is used", file=sys.stderr)
    n_gpus = world
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import vuke_amd
    from vuke_amd import _ffi as ffi
    import bench_workloads as wl

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t_wall0 = time.perf_counter()
    try:
        eng = vuke_amd.Engine(local_rank, window_bits=args.window_bits, batch=args.batch, blocks_per_cu=args.blocks_per_cu, layout=args.layout)
    except vuke_amd.VkError as exc:
        if args.layout == 0 or exc.code not in (ffi.VK_E_NOMEM, ffi.VK_E_HIP):   # NOMEM since round 6 (hipErrorOutOfMemory); HIP from older builds
            raise
        # the 120 GB table did not fit (shared or smaller device): fall back to the library's default geometry
        print(f"bench.py: layout {args.layout} unavailable ({exc}); using the default 24-bit uniform table", file=sys.stderr)
        eng = vuke_amd.Engine(local_rank, batch=args.batch, blocks_per_cu=args.blocks_per_cu)
    engine_create_s = time.perf_counter() - t_wall0
    info = eng.info()

    w = make_workload(args, rank, world)
    w.setup(eng, ffi, wl)
    if args.strong and args.emulate_gpus:
        w.per_rank = w.nblocks // args.emulate_gpus
    setup_s = time.perf_counter() - t_wall0 - engine_create_s
    steps = w.step_total()

    for _ in range(args.warmup):
        w.warmup(eng)

    units = keys_done = matches = expected = launches = 0
    kernel_s = clock_acc = 0.0
    digest = (0, 0, 0, 0)
    full_after = w.full_sweep_steps() if (args.config == "2" and not args.emulate_gpus) else 0
    t_full = 0.0
    step_ms = []
    import gc
    gc.collect()
    gc.disable()   # no collector pause inside the timed region (the steps allocate only small per-step records)
    sync()
    t0 = time.perf_counter()
    t_prev = t0
    for s in range(steps):
        st = w.step(eng, s)
        t_now = time.perf_counter()
        step_ms.append(1e3 * (t_now - t_prev))   # vk_sweep / vk_scan_strings return after the device work of the step is done
        t_prev = t_now
        if s + 1 == full_after:   # vk_sweep has returned its counters: this rank's share of one whole 2^32 sweep is done
            t_full = time.perf_counter() - t0
        units += st["units"]
        keys_done += st["keys"]
        matches += st["matches"]
        expected += st["expected"]
        kernel_s += st["kernel_seconds"]
        clock_acc += st.get("shader_clock_ghz", 0.0) * st["kernel_seconds"]
        launches += st["launches"]
        digest = wl.fold_digests([digest, st["digest"]])
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()

    # counters: one tiny all-reduce over RCCL (the only collective on this path)
    red_dev = dev if (not use_dist or args.backend == "nccl") else torch.device("cpu")
    (total_units, total_keys, total_matches, total_expected, dg_checked, dg_bad), (t, t_full, create_max, setup_max) = reduce_counters(
        dist, red_dev, [units, keys_done, matches, expected, w.digest_checked, w.digest_bad], [elapsed, t_full, engine_create_s, setup_s])
    wall_s = time.perf_counter() - t_wall0
    # the digests of all ranks (XOR words and the wrapping sum): gathered, folded on rank 0
    all_digests = [digest]
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, [int(x) for x in digest])
        all_digests = [tuple(g) for g in gathered]
    if total_matches != total_expected:
        print(f"bench.py: PARITY FAILURE: {total_matches} matches, {total_expected} planted in range", file=sys.stderr)
        sys.exit(3)
    if dg_bad:
        print(f"bench.py: PARITY FAILURE: {dg_bad} of {dg_checked} digests differ from the oracle's (tests/golden)", file=sys.stderr)
        sys.exit(3)

    # ---- one whole 2^32 sweep as the product would run it when that is the whole job (outside the timed region: `value` is done)
    strong_after = None
    eng_big = None
    if args.config == "2" and not args.strong and not args.no_strong_after and not args.emulate_gpus and w.nblocks % world == 0 and world <= w.nblocks:
        import copy
        # the big engine stays alive until the end: 124 GB + 47 GB (or 14 GB) fit the 288 GB part, and memory this process has just
        # released would have to be scrubbed by the driver before it could be handed out again (2 - 6 s for 124 GB)
        eng_big = eng
        a2 = copy.copy(args)
        a2.strong = True
        sync()
        ts0 = time.perf_counter()
        # by measured wall time (profiles/archive/r02_strong_geometry.txt) the big table cannot pay for itself inside one sweep: ten signed
        # windows on one GPU (18 GB: the rate of the unsigned 26-bit table at 0.4 of its size, profiles/archive/r04_signed_geometries.txt),
        # eleven (4 GB) from two GPUs
        g_bits = 26 if world == 1 else 24
        try:
            eng = vuke_amd.Engine(local_rank, window_bits=g_bits, batch=args.batch, blocks_per_cu=args.blocks_per_cu, layout=1)
            created = 1
        except vuke_amd.VkError as exc:   # the headline is already measured: a rank that cannot get the second table must not take the line down
            print(f"bench.py: rank {rank}: no engine for the one-sweep run ({exc}); skipping it", file=sys.stderr)
            eng, created = None, 0
        (n_created,), _ = reduce_counters(dist, red_dev, [created], 0.0)
        if n_created != world:
            if eng is not None:
                eng.close()
            eng = None
        s_create = time.perf_counter() - ts0
    if args.config == "2" and not args.strong and not args.no_strong_after and not args.emulate_gpus and w.nblocks % world == 0 and world <= w.nblocks and eng is not None:
        s_info = eng.info()
        w2 = Milksad(a2, rank, world)
        w2.setup(eng, ffi, wl)
        s_setup = time.perf_counter() - ts0 - s_create
        s_keys = s_matches = s_expected = 0
        s_digest = (0, 0, 0, 0)
        ts1 = time.perf_counter()
        for s in range(w2.step_total()):
            st = w2.step(eng, s)
            s_keys += st["keys"]
            s_matches += st["matches"]
            s_expected += st["expected"]
            s_digest = wl.fold_digests([s_digest, st["digest"]])
        sync()
        s_sweep = time.perf_counter() - ts1
        (sk, sm, se, sdc, sdb), (s_sweep_max, s_create_max, s_setup_max) = reduce_counters(
            dist, red_dev, [s_keys, s_matches, s_expected, w2.digest_checked, w2.digest_bad], [s_sweep, s_create, s_setup])
        s_wall = time.perf_counter() - ts0
        s_digests = [s_digest]
        if use_dist:
            gathered = [None] * world
            dist.all_gather_object(gathered, [int(x) for x in s_digest])
            s_digests = [tuple(g) for g in gathered]
        if sm != se or sdb:
            print(f"bench.py: PARITY FAILURE in the one-sweep run: {sm} matches / {se} planted, {sdb} of {sdc} digests differ", file=sys.stderr)
            sys.exit(3)
        strong_after = {"covers_full_2p32": sk == 1 << 32, "keys": sk, "blocks_per_rank": w2.step_total(), "layout": s_info["layout"],
                        "window_bits": s_info["window_bits"], "windows": s_info["windows"], "table_gb": s_info["table_bytes"] / 1e9, "engine_create_s": s_create_max,
                        "engine_create_split_rank0": {"hip_runtime_s": s_info["runtime_init_seconds"], "alloc_s": s_info["alloc_seconds"],
                                                      "table_build_s": s_info["table_build_seconds"], "table_wall_s": s_info["table_wall_seconds"]},
                        "targets_and_fixtures_s": s_setup_max, "sweep_s": s_sweep_max, "wall_s": s_wall, "keys_per_s_end_to_end": sk / s_wall,
                        "digest": wl.hexd(wl.fold_digests(s_digests)), "digests_checked_against_oracle": sdc, "matches": sm,
                        "note": "measured AFTER the timed region, on the table geometry the product picks for ONE sweep; wall_s = from before "
                                "vk_engine_create to after the counter all-reduce (max over ranks by construction: barriers on both sides)"}

    if rank == 0:
        value = total_keys / t
        per_launch_keys = keys_done / max(launches, 1)
        per_launch_s = kernel_s / max(launches, 1)
        rate = keys_done / kernel_s  # this rank's keys per second of kernel time
        pmc, why = load_pmc(w.tag, w.kernel)
        roof = {"bound": "valu-int", "peak": PEAK_INT32 / 1e12,
                "unit": "T lane-ops/s = executed VALU wave-instructions (rocprofv3 SQ_INSTS_VALU) x 64 lanes / s; peak = the 16-lanes-per-clock "
                        "class (v_mad_u64_u32, carry adds, v_add3, v_alignbit): 256 CU x 4 SIMD x 16 lanes x 2.4 GHz",
                "kernel": w.kernel, "kernel_ms_per_launch": 1e3 * per_launch_s, "keys_per_launch": per_launch_keys}
        clock_ghz = clock_acc / kernel_s if kernel_s else 0.0
        if clock_ghz:
            roof["shader_clock_ghz"] = clock_ghz  # live: s_memtime / s_memrealtime of workgroup 0 of every sweep launch of this run
        if pmc:
            per_key = pmc["valu_insts_per_key"]
            floor = pmc.get("cycles_per_valu_inst_floor")
            if clock_ghz and floor:
                # at the clock the part actually held (DVFS lowers it under this load): the priced issue floor of this kernel's own
                # instruction mix at its own occupancy (profiles/valu_rates.json) over the cycles per instruction of THIS run --
                # <= 1 by construction of the floor; what is missing to 1 is waiting (wave_cycle_split below)
                live_cycles = clock_ghz * 1e9 * 1024 * 64 / (per_key * rate)
                roof["cycles_per_valu_inst_per_simd_live"] = live_cycles
                roof["frac_at_measured_clock"] = floor / live_cycles
            roof.update({"achieved": per_key * rate / 1e12, "frac": per_key * rate / PEAK_INT32,
                         "valu_insts_per_key": per_key, "pmc_source": pmc["source"], "pmc_csrc_hash": pmc["csrc_hash"], "pmc_git": pmc.get("git"),
                         "cycles_per_valu_inst_per_simd": pmc.get("cycles_per_valu_inst_per_simd"),
                         "valu_busy_pct_in_profile": pmc.get("valu_busy_pct"),
                         "effective_clock_ghz_in_profile": pmc.get("effective_clock_ghz"),
                         "traffic": pmc.get("hbm_bytes_per_launch_calibrated", pmc.get("hbm_bytes_per_launch")),
                         "cycles_per_valu_inst_floor": pmc.get("cycles_per_valu_inst_floor"), "wave_cycle_split_in_profile": pmc.get("wave_cycle_split"),
                         "valu_int64_share": (pmc.get("valu_mix") or {}).get("int64_share"),
                         "scratch": pmc.get("dispatch", {}).get("Scratch_Size")})
        else:
            roof.update({"achieved": None, "frac": None, "traffic": None, "pmc_refused": why})
        km = kernel_meta(w.kernel)
        if km:   # registers the CODE OBJECT allocates per lane (tools/kernel_meta.py), not rocprof's dispatch field (which counts in units of two)
            roof.update({"vgpr": km["vgpr"], "agpr": km["agpr"], "vgpr_spills": km["vgpr_spill_count"], "waves_per_simd": max(1, 512 // max(km["vgpr"] + km["agpr"], 1)),
                         "static_cycles_per_valu_inst_floor": km.get("cycles_per_valu_inst_floor")})
        if hasattr(w, "canonical_ops_per_key"):
            roof["canonical_ops_per_key"] = w.canonical_ops_per_key
            roof["canonical_frac"] = w.canonical_ops_per_key * rate / PEAK_INT32
        else:
            roof["canonical_ops_per_input"] = w.canonical_ops_per_input
            roof["canonical_frac"] = w.canonical_ops_per_input * (units / kernel_s) / PEAK_INT32
        roof["canonical_note"] = ("SURVEY 8(d)'s fixed 4-bit-window reference algorithm; the shipped kernel does far fewer operations per key "
                                  "(9 windows in a 120 GB HBM table), so this ratio exceeds 1 and is NOT a utilisation figure -- frac is")
        out = {
            "metric": "keys/sec (seed->P2PKH addr)" + (", Milksad sweep" if args.config == "2" else f", BASELINE config {args.config}"),
            "value": value,
            "unit": "keys/s",
            "n_gpus": n_gpus,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * t / steps,
            "ms_per_step_rank0": {"min": min(step_ms), "median": sorted(step_ms)[len(step_ms) // 2], "max": max(step_ms), "argmax": step_ms.index(max(step_ms))},
            "higher_is_better": True,
            "scaling": "strong" if (args.strong or args.config in ("5m", "5e")) else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": w.describe(eng),
                "baseline_config": int(args.config[0]),
                "window_bits": info["window_bits"], "windows": info["windows"], "layout": info["layout"],
                "table_gb": info["table_bytes"] / 1e9, "table_build_seconds": info["table_build_seconds"],
                "batch": info["batch"], "parallelism": f"range-partition x{n_gpus}" if args.config in ("2", "3", "4") else f"line-partition x{n_gpus}",
            },
            "matches": total_matches,
            "keys_total": total_keys, "units_total": total_units,
            "digest": wl.hexd(wl.fold_digests(all_digests)),
            "digest_ok": (dg_bad == 0) if dg_checked else None,
            "digests_checked_against_oracle": dg_checked,
            "roofline": roof,
        }
        if w.unit_name == "inputs":
            out["inputs_per_s"] = total_units / t
        # what the job pays before the first key: slowest rank's vk_engine_create (allocation + table build on the device) and
        # target upload; outside `value` (sustained throughput), inside every *_end_to_end_s below
        out["engine_create_s"] = create_max
        out["table_build_s"] = info["table_build_seconds"]
        # where rank 0's vk_engine_create went (vk_engine_info, ABI 5): HIP runtime start, the two large hipMallocs (this is where a
        # box waits while the driver scrubs VRAM a previous process has just freed), the table kernels (hipEvent time / host wall)
        out["engine_create_split_rank0"] = {"total_s": engine_create_s, "hip_runtime_s": info["runtime_init_seconds"], "alloc_s": info["alloc_seconds"],
                                            "table_build_s": info["table_build_seconds"], "table_wall_s": info["table_wall_seconds"],
                                            "other_s": engine_create_s - info["runtime_init_seconds"] - info["alloc_seconds"] - info["table_wall_seconds"]}
        out["targets_and_fixtures_s"] = setup_max
        if args.config == "2":
            out["sweep_2p32_seconds_at_this_rate"] = (1 << 32) / value
            if full_after:
                # MEASURED, not extrapolated: the first 16/N timed steps of the N ranks are disjoint blocks covering the whole seed
                # space (block_of), every block's digest checked against the oracle's; max over ranks of the time they took
                out["sweep_2p32_wall_s"] = t_full
                out["sweep_2p32_steps"] = full_after
                out["sweep_2p32_end_to_end_s"] = create_max + setup_max + t_full
            ab = w.algo_bytes_per_key(info)
            out["roofline_hbm"] = {"bound": "hbm", "achieved": ab * rate / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ab * rate / HBM_PEAK,
                                   "traffic": roof.get("traffic"), "algorithmic_bytes_per_key": ab,
                                   "table_gathers_per_s": info["windows"] * rate,
                                   "table_gather_ceiling_per_s": 48.5e9}  # quad-cooperative 64-B reads landing in LDS, tools/lds_gather_test.hip
        if strong_after:
            out["strong"] = strong_after
        if args.strong:
            covered = steps * world == w.nblocks and not args.emulate_gpus
            out["strong"] = {"blocks_per_rank": steps, "covers_full_2p32": covered, "engine_create_s": engine_create_s,
                             "table_build_s": info["table_build_seconds"], "targets_and_fixtures_s": setup_s, "sweep_s": t, "wall_s": wall_s,
                             "emulated_gpus": args.emulate_gpus or None,
                             "note": "wall_s = from before vk_engine_create (table allocation + build) to after the counter all-reduce, warm-up included"}
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(out), flush=True)

    if eng is not None:
        eng.close()
    if eng_big is not None and eng_big is not eng:
        eng_big.close()
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(config):
    """The CPU oracle ("port": our C restatement, not the Rust binary) on the host cores, on a bounded sample of the same
    workload.  value = the leanest form (one k*G per key through 16-bit windows and a shared inversion + only the hash160s
    the config needs); ref_faithful = everything KeyDeriver::derive does per key."""
    from oracle import vko
    import bench_workloads as wl
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("VUKE_CPU_THREADS", "16"))))  # the 1-GPU box's CPU share is 16

    def timed(sp, n, flags):
        t = time.perf_counter()
        st, _ = vko.sweep(sp, 0, n - 1, flags=flags, nthreads=cores)
        return st["keys"] / (time.perf_counter() - t), time.perf_counter() - t

    if config in ("2", "3", "4"):
        kind, fl, n = {"2": (vko.T_MILKSAD, 1, 1 << 25), "3": (vko.T_SHA256, 3, 1 << 23), "4": (vko.T_LCG, 1, 1 << 23)}[config]
        sp = vko.spec(kind)
        vko.sweep(sp, 0, 4095, flags=fl | vko.F_BATCHED, nthreads=cores)  # builds both tables, warms up
        v, dt = timed(sp, n, fl | vko.F_BATCHED)
        vu, dtu = timed(sp, n >> 3, fl)
        vf, dtf = timed(sp, n >> 4, 3 | vko.F_FAITHFUL)
        return {"value": v, "unit": "keys/s", "cores": cores, "kind": "port",
                "sample": f"inputs [0, {n}) of the same workload, oracle/ C restatement with batched k*G (16-bit windows, one inversion per 1000-input chunk), {cores} threads, {dt:.1f} s",
                "unbatched_value": vu, "unbatched_sample": f"inputs [0, {n >> 3}), 8-bit windows + one inversion per key, {dtu:.1f} s",
                "ref_faithful_value": vf,
                "ref_faithful_sample": f"inputs [0, {n >> 4}), everything KeyDeriver::derive does per key (3 x k*G, both hash160, base58/bech32/WIF/decimal strings), {dtf:.1f} s"}
    # config 5: the KDF chain dominates; the oracle's chain functions over a thread pool (ctypes releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    if config == "5m":
        words = vko.Wordlist(wl.c5_wordlist())
        lines = wl.c5_mnemonics(6000)
        fn = lambda m: vko.multibit_keys(words, m)
    else:
        lines = wl.c5_electrum_seeds(1000)
        fn = lambda s: vko.electrum_keys(s)
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(fn, lines[:cores]))
        t = time.perf_counter()
        res = list(ex.map(fn, lines))
        keys = 0
        for r in res:
            ks = r[2] if config == "5m" else r[1]
            keys += len(ks)
        for r in res[:64]:  # derive share (small next to the chain): timed on a slice, scaled
            for _, k in (r[2] if config == "5m" else r[1]):
                vko.derive_h160(k)
        dt = time.perf_counter() - t
    return {"value": keys / dt, "unit": "keys/s", "inputs_per_s": len(lines) / dt, "cores": cores, "kind": "port",
            "sample": f"the first {len(lines)} inputs of the same workload through the oracle's chain (python thread pool over the C functions), {cores} threads, {dt:.1f} s"}


if __name__ == "__main__":
    main()
