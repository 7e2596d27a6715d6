#!/usr/bin/env python3
"""Summary of a tools/pmc_forms.sh run: per case and sweep kernel, every counter per key (the probe's blocks are 2^28 keys for
config 2 / config 4, 3 x 2^25 for config 3, 2^27 for the Milksad scan of both key forms; the untimed first block counts too:
two launches per kernel).   tools/pmc_forms.py <outdir>"""
import collections, csv, glob, os, sys
root = sys.argv[1]
for case in sorted(os.listdir(root)):
    d = os.path.join(root, case)
    if not os.path.isdir(d):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
        first = "p1" + os.sep in f
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "vk_sweep_kernel" not in k:
                continue
            k = k.split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if first and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                launches[k] += 1
            for key in ("VGPR_Count", "Accum_VGPR_Count", "Scratch_Size", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                if key in r:
                    acc[k]["_" + key] = float(r[key])
    for k, c in acc.items():
        keys = {"<1u": 2 ** 28, "<3u": 2 ** 28, "<0u": 3 * 2 ** 25}.get(k[k.index("<"):k.index("<") + 3], 0)
        if os.environ.get("PMC_FORMS_MORE_" + case.upper()) or "more" in case:
            keys = {"<1u": 2 ** 27 , "<3u": 2 ** 28}.get(k[k.index("<"):k.index("<") + 3], keys)
        n = max(1, launches[k]) * keys
        print(f"== {case}  {k}  launches={launches[k]} keys/launch={keys}")
        for name in sorted(c):
            v = c[name]
            print(f"   {name:24s} {v:16.0f}" + ("" if name.startswith("_") or not n else f"   per key {v / n:10.3f}"))
        if c.get("SQ_INSTS_VALU") and c.get("GRBM_GUI_ACTIVE"):
            print(f"   cycles per VALU inst per SIMD  {c['GRBM_GUI_ACTIVE'] * 1024 / c['SQ_INSTS_VALU']:.3f}   (GRBM_GUI_ACTIVE x 1024 SIMDs / SQ_INSTS_VALU)")
        if c.get("SQ_WAVE_CYCLES"):
            for w in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS"):
                if w in c:
                    print(f"   {w} / SQ_WAVE_CYCLES = {c[w] / c['SQ_WAVE_CYCLES']:.4f}")
