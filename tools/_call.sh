set -x
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -40
python - <<'PY'
import json
r=json.load(open('gpurun_out/parity_report.json'))
for k,v in r.items():
    if 'full_micro_step' in k: print(k, max(x['measured'] for x in v), [round(x['measured'],4) for x in v])
PY
