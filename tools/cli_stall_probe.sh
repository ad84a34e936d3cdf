# whole-assembly command-line runs under a few runtime settings: which of them shows the tens-of-ms stalls
cd "$GRAFT_REPO_ROOT"
for v in "X=1" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=8" "HIP_FORCE_DEV_KERNARG=1" "NP2_IO_THREADS=8" "AMD_SERIALIZE_KERNEL=0 HSA_ENABLE_SDMA=0"; do
  echo "== $v"
  env $v NP2_CLI_PROFILE=1 timeout 300 python tools/cli_probe.py > gpurun_out/stall.log 2>&1
  grep -E "^-t|front end [0-9]{2}\.|polish [0-9]{2}\." gpurun_out/stall.log | sed -e 's/; host clock.*//' | tail -n +8
done
