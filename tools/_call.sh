timeout 600 python -m pytest tests/test_vae_gpu.py -m gpu -x -q -k batch_of_two 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
