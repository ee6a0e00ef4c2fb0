rm -f gpurun_out/parity_report.json
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
