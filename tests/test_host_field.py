"""The generated field arithmetic on a machine without a GPU: vuke_amd/csrc/vk_field.cuh compiled for the host with its
primitive and generated headers swapped for tests/host_emu/ -- tools/gen_field.py writes every inline-asm statement a second time
as calls into a one-lane instruction emulator (vk_field_gen_host.h, vk_field_prims_host.h) -- and held to python integers on the operands of the device's field test (every pair of edge values,
all aliasing forms, the flagged and the exact reduction).  This checks the ARITHMETIC of the generated code (carry accounting,
rare paths, operand order); hardware hazards and the compiler are what the GPU tests and tools/lint_coalescer.py check."""
import filecmp
import os
import subprocess
import sys

import pytest

from tests.test_gpu_parity import P, field_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vuke_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "host_emu")


def build(tmp_path, *defines):
    exe = tmp_path / ("emu" + "".join(defines).replace("=", "_").replace("-D", "_"))
    cmd = ["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas", "-I", CSRC, "-I", EMU, *defines, "-o", str(exe), os.path.join(ROOT, "tests", "host_emu", "host_field_emu.cpp")]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return str(exe)


def run(exe, op, a, b):
    inp = "".join(f"{op} {x:064x} {y:064x}\n" for x, y in zip(a, b))
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split()
    assert len(out) == len(a)
    return [int(h, 16) for h in out]


def test_generated_files_are_what_the_generator_writes(tmp_path):
    env = dict(os.environ, VK_GEN_OUT_DIR=str(tmp_path))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_field.py")], check=True, env=env, capture_output=True)
    for d, f in ((CSRC, "vk_field_gen.cuh"), (EMU, "vk_field_gen_host.h")):
        assert filecmp.cmp(os.path.join(d, f), str(tmp_path / f), shallow=False), f"{f} is stale: run tools/gen_field.py"
    assert "asm(" not in open(os.path.join(EMU, "vk_field_gen_host.h")).read()
    # test infrastructure stays out of the kernel sources (and so out of the hash the PMC profiles are keyed on)
    assert not os.path.exists(os.path.join(CSRC, "vk_field_gen_host.h"))
    assert "HOST_EMU" not in open(os.path.join(CSRC, "vk_field.cuh")).read()


@pytest.mark.parametrize("defines", [(), ("-DVK_REDUCE_FLAGS=0",)], ids=["flagged-reduction", "round1-reduction"])
def test_host_emulation_of_the_generated_field_code(tmp_path, defines):
    exe = build(tmp_path, *defines)
    ops, a, b = field_cases()
    for op, f in ops + [(100, lambda x, y: x * y % P), (101, lambda x, y: x * x % P), (102, lambda x, y: 2 * x * y % P)]:   # 100 - 102: the exact product paths alone
        got = run(exe, op, a, b)
        for x, y, g in zip(a, b, got):
            assert g == f(x, y), (op, hex(x), hex(y))
    nz = [x for x in a if x % P][:120]
    for x, g in zip(nz, run(exe, 2, nz, nz)):
        assert g == pow(x, -1, P)


def test_host_emulation_sees_a_dropped_carry(tmp_path):
    """The check has teeth: with the recomputation of flagged products compiled out (VK_TIMING_ONLY_NO_RECOMPUTE, the round-2
    timing experiment) the edge operands must produce wrong products."""
    exe = build(tmp_path, "-DVK_TIMING_ONLY_NO_RECOMPUTE=1")
    _, a, b = field_cases()
    got = run(exe, 0, a, b)
    assert any(g != x * y % P for x, y, g in zip(a, b, got))
    got = run(exe, 16, a, b)   # the fused pair of products has flags of its own (limb 16, the second short chain)
    assert any(g != 2 * x * y % P for x, y, g in zip(a, b, got))
