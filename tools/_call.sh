set -x
timeout 800 bash tools/ab_gemm_traffic.sh "1 4 8" 2>&1 | tail -150
