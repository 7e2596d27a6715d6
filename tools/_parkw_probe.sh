#!/bin/bash
# wide form with the running product parked in LDS (variant) against the shipped wide form: timing (interleaved), then counters per key
set -uo pipefail
export TMPDIR=/tmp
V=vuke_amd/lib/variants/libvuke_hip_parkw.so
for r in 1 2 3; do
  python3 tools/probe_split.py 3 0
  VUKE_HIP_LIB=$V python3 tools/probe_split.py 3 0
done
bash tools/pmc_forms.sh gpurun_out/r06p/pmc "ship:-:0" "parkw:$V:0" && python3 tools/pmc_forms.py gpurun_out/r06p/pmc
