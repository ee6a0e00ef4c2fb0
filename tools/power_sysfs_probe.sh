#!/bin/bash
# What the GPU box exposes for socket power / shader clock (feeds bench.py's PowerSampler): run once per round, output is a log, not a record.
for c in /sys/class/drm/card[0-9]*/device; do
  echo "== $c -> $(realpath $c)"
  for h in $c/hwmon/hwmon*; do
    for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq1_label; do [ -e $h/$f ] && echo "$h/$f: $(cat $h/$f 2>&1)"; done
  done
  [ -e $c/pp_dpm_sclk ] && { echo "pp_dpm_sclk:"; cat $c/pp_dpm_sclk; }
done
python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print("torch device 0:", p.name, getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None))
PY
rocm-smi --showpower --showclocks 2>&1 | head -20
