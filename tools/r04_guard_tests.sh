#!/bin/bash
# Round 4: the whole GPU test suite under the guard-page allocator (one process, -x), then the opt-in MIOpen solver search
# of the attack group under it (expected to fault inside MIOpen's igemm_bwd candidate - see DESIGN.md section 6).
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
export NERFAIL_GUARD_ALLOC=1
timeout -k 10 ${T:-1000} python3 -m pytest tests -m gpu -x -v -p no:cacheprovider > $O/guard_pytest.log 2>&1
rc=$?
echo "pytest under guard rc $rc" | tee $O/guard_pytest.status
tail -n 15 $O/guard_pytest.log
[ $rc -ne 0 ] && exit $rc
if [ "${TUNED:-1}" = "1" ]; then
  export NERFAIL_BENCH_TUNE_VICTIM=1 HIP_LAUNCH_BLOCKING=1 AMD_LOG_LEVEL=3
  timeout -k 10 600 python3 bench.py --child attack > $O/guard_tuned.out 2> $O/guard_tuned.err
  rc=$?
  echo "tuned attack under guard rc $rc" | tee -a $O/guard_pytest.status
  grep "ShaderName\|fault\|exception" $O/guard_tuned.err | tail -n 12 | cut -c1-260 > $O/guard_tuned.kernels
  rm -f $O/guard_tuned.err
  cat $O/guard_tuned.kernels | tail -4
fi
exit 0
