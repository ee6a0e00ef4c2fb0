# tools/rep_matrix.sh: share_probe gemmqk under 6-process sharing, per library variant (TG_LIB_PATH); prints differing launches per variant
run6() { label=$1; lib=$2; n=$3
  for i in 1 2 3 4 5; do TG_LIB_PATH=$lib python tools/share_probe.py gemmqk $n $label$i > gpurun_out/sp_${label}_$i.log 2>&1 & done
  TG_LIB_PATH=$lib python tools/share_probe.py gemmqk $n ${label}0 > gpurun_out/sp_${label}_0.log 2>&1
  wait
  echo "== $label: $(cat gpurun_out/sp_${label}_*.log | grep -c 'differ, first') differing launches of $((6*n)) ($(cat gpurun_out/sp_${label}_*.log | grep -c SHARE_PROBE) processes finished)"
}
V=$PWD/tokensgen_amd/csrc/variants
N=${1:-300000}
run6 product "" $N
run6 v1_uncond $V/qkv1.so $N
run6 v2_wait $V/qkv2.so $N
run6 v3_noslp $V/qkv3.so $N
run6 product2 "" $N
