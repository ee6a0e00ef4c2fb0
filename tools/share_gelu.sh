# tools/share_gelu.sh [N]: the GELU-epilogue GEMMs under 6-process GPU sharing (tools/share_probe.py gemmgelu / gemmkeep); prints differing launches per mode
N=${1:-300000}
mkdir -p gpurun_out
for mode in gemmgelu gemmkeep; do
  for i in 1 2 3 4 5; do timeout 600 python tools/share_probe.py $mode $N ${mode}$i > gpurun_out/sg_${mode}_$i.log 2>&1 & done
  timeout 600 python tools/share_probe.py $mode $N ${mode}0 > gpurun_out/sg_${mode}_0.log 2>&1
  wait
  echo "== $mode: $(cat gpurun_out/sg_${mode}_*.log | grep -c 'differ, first') differing launches of $((6*N)) ($(cat gpurun_out/sg_${mode}_*.log | grep -c SHARE_PROBE) of 6 processes finished)"
  grep -h "differ, first" gpurun_out/sg_${mode}_*.log | head -5
  grep -h "Error\|Traceback" gpurun_out/sg_${mode}_*.log | head -3
done
