export TMPDIR=/tmp
rm -f gpurun_out/parity_report.json
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
TG_ATTN_BWD_V1=1 timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -q -k "attention_bwd_vs or processor_attention" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -1 gpurun_out/bench_final.json | cut -c1-2500
