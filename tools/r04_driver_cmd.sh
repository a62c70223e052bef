#!/bin/bash
# Round 4: the driver's exact bench command, N times in a row, each line kept.
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
N=${N:-2}
for i in $(seq 1 $N); do
  k=$(ls $O/driver_cmd_*.json 2>/dev/null | wc -l); k=$((k+1))
  t0=$SECONDS
  timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_$k.json 2> $O/driver_cmd_$k.err
  rc=$?
  echo "run $k rc $rc wall $((SECONDS-t0)) s bytes $(wc -c < $O/driver_cmd_$k.json)" | tee -a $O/driver_cmd.status
  tail -n 5 $O/driver_cmd_$k.err > $O/driver_cmd_$k.err.tail; rm -f $O/driver_cmd_$k.err
  [ $rc -ne 0 ] && exit $rc
done
exit 0
