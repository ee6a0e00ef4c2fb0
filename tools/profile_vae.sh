#!/bin/bash
# rocprofv3 evidence for the VAE (BASELINE config 4): kernel-trace stats, then PMC passes in their own runs (no trace domains).
# Usage (GPU box, repo root): tools/profile_vae.sh TAG [pmc]  -> gpurun_out/prof_vae_TAG/
set -u
tag=${1:-r2}
R=$PWD
out=$R/gpurun_out/prof_vae_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python $R/tools/bench_vae.py --plain > $out/bench_under_rocprof.json 2> $out/stats.err
if [ "${2:-}" = "pmc" ]; then
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -o pmc -- python $R/tools/bench_vae.py --plain decode > /dev/null 2> $out/pmc1.err
  rocprofv3 --pmc WRITE_SIZE SQ_BUSY_CYCLES --output-format csv -d $out/pmc2 -o pmc -- python $R/tools/bench_vae.py --plain decode > /dev/null 2> $out/pmc2.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc3 -o pmc -- python $R/tools/bench_vae.py --plain decode > /dev/null 2> $out/pmc3.err
fi
cd $R
find $out -name "*kernel_trace.csv" -delete
ls -la $out/*
