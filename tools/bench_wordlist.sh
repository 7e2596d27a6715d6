#!/bin/bash
# passphrase-source throughput through the CLI (host file parsing + vk_transform_strings + vk_scan_keys)
set -euo pipefail
n="${1:-20000000}"
python3 - "$n" <<'PY'
import sys, random
n = int(sys.argv[1])
rnd = random.Random(1)
words = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel", "india", "juliet", "kilo", "lima"]
with open("/tmp/wl.txt", "w") as f:
    for i in range(n):
        f.write(f"{words[i % 12]} {words[(i * 7) % 12]} {i}\n")
    f.write("correct horse battery staple\n")
open("/tmp/t.txt", "w").write("1JwSSubhmg6iPtRjtyqhUYYH7bZg3Lfy1T\n")
PY
ls -la /tmp/wl.txt
for t in "sha256" "sha256 double_sha256 md5"; do
  echo "== $t"
  time ./vuke_amd/lib/vuke-hip scan --transform $t --targets /tmp/t.txt wordlist --file /tmp/wl.txt > /tmp/out.txt
  grep -c HIT /tmp/out.txt || true
done
