#!/usr/bin/env python3
"""Issue cost per VALU instruction class and occupancy, from the micro-benchmarks' own output:

    tools/ubench > a.txt; tools/ubench_carry > b.txt; python tools/ubench_rates.py a.txt b.txt profiles/valu_rates.json

For every class the static / dynamic instruction mix of a kernel is priced with (tools/kernel_meta.py, tools/pmc_summary.py) and for
every measured number of resident waves per SIMD: the SMALLEST wall-clock cost (shader cycles per wave-instruction per SIMD) any
member of the class reached, and `best_any_occupancy`: the smallest over all occupancies = what the pipe sustains for that class
when the SIMD has enough waves to choose from.  The FLOOR a kernel's mix is held against is priced with the latter (`measured
cycles per instruction >= floor`; bench.py refuses a summary that violates it); the price at the kernel's own occupancy is
reported beside it but is NOT a floor: a real mix at three waves issues ~2 % faster than the sum of its classes' single-class
streams at three waves (waves in different classes at the same moment overlap better than three waves of one class).  The multiplier class also takes the multiply-add / carry
patterns of tools/ubench_carry.hip (several instructions per asm block: no s_nop between them, as in the shipped field layer)."""
import json
import re
import sys

CLASSES = {
    "mad64": ["v_mad_u64_u32"],
    "full": ["v_add3_u32", "v_bfi_b32", "v_alignbit_b32", "v_addc_co_u32", "v_lshl_add_u32", "v_xad_u32", "v_perm_b32", "v_mul_lo_u32", "v_mul_hi_u32",
             "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_lshl_add_u64", "v_cmp_lt_u32", "v_addc_co_u32 (sgpr carry)"],
    "bitop3": ["v_bitop3_b32"],
    "mov": ["v_mov_b32"],
    "fast": ["v_add_u32", "v_xor_b32"],
}


def main():
    a, b, out = sys.argv[1], sys.argv[2], sys.argv[3]
    per = {}   # waves -> instruction -> wall cycles
    for line in open(a):
        m = re.match(r"^(v_\S+(?: \([^)]*\))?)\s+waves/SIMD=(\d+) live=64\s+[\d.]+ cyc/wave-instr/SIMD \(wall\s+([\d.]+)\)", line)
        if m:
            per.setdefault(int(m.group(2)), {}).setdefault(m.group(1), []).append(float(m.group(3)))
    carry = {}
    for line in open(b):
        m = re.match(r"^(\w+)\s+\(.*\)\s+waves/SIMD=(\d+)\s+one wave:.*SIMD:\s+([\d.]+) cycles per instruction", line)
        if m:
            carry.setdefault(int(m.group(2)), {})[m.group(1)] = float(m.group(3))
    rates = {}
    for w, ins in sorted(per.items()):
        row = {}
        for cls, members in CLASSES.items():
            vals = [min(ins[x]) for x in members if x in ins]
            if cls == "mad64":
                vals += list(carry.get(w, {}).values())
            if vals:
                row[cls] = min(vals)
        rates[str(w)] = row
    best = {cls: min(row[cls] for row in rates.values() if cls in row) for cls in CLASSES}
    json.dump({"tool": "tools/ubench_rates.py", "best_any_occupancy": best, "sources": [a, b], "unit": "shader cycles per wave-instruction per SIMD (wall), smallest member of the class",
               "classes": CLASSES, "waves_per_simd": rates, "per_instruction": {str(w): {k: min(v) for k, v in ins.items()} for w, ins in sorted(per.items())},
               "multiplier_patterns": {str(w): v for w, v in sorted(carry.items())}}, open(out, "w"), indent=1)
    for w, row in rates.items():
        print(w, row)
    print("best", best)


if __name__ == "__main__":
    main()
