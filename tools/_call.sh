set -x
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "model_training or block_backward" 2>&1 | tail -40
python - <<'PY'
import json
r=json.load(open('gpurun_out/parity_report.json'))
for k,v in r.items():
    if 'model_training' in k: print(k, [round(x['measured'],5) for x in v])
PY
