set -e
mkdir -p gpurun_out/r06f
O=gpurun_out/r06f/ab.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_kind or priority or ragged or allocation" > gpurun_out/r06f/parity.log 2>&1 || { tail -40 gpurun_out/r06f/parity.log; exit 1; }
tail -2 gpurun_out/r06f/parity.log
for r in 1 2 3; do
  for lib in vuke_amd/lib/variants/libvuke_hip_d24.so vuke_amd/lib/libvuke_hip.so; do
    for cfg in 2 3 4; do
      echo "lib=$lib cfg=$cfg" >> $O
      VUKE_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-strong-after --config $cfg --steps 6 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'M keys/s', round(d['ms_per_step'],2), 'ms/step', d.get('roofline',{}).get('shader_clock_ghz'))" >> $O
    done
  done
done
cat $O
