timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k full_width 2>&1 | tail -12
python - <<'PY'
import json
r=json.load(open('gpurun_out/parity_report.json'))
for k,v in r.items():
    if 'full_width_block_backward' in k: print(max(x['measured'] for x in v), [round(x['measured'],4) for x in v])
PY
