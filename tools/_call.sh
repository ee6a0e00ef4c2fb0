export TMPDIR=/tmp
timeout 120 python tools/attn_bwd_check.py 2>&1 | grep "dq"
for v in base noprio; do
  lib=$PWD/tokensgen_amd/csrc/variants/$v.so; [ $v = base ] && lib=$PWD/tokensgen_amd/libtokensgen_hip.so
(cd /tmp; rm -rf /tmp/p_x; TG_LIB_PATH=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_x -o stats -- python $OLDPWD/tools/bench_kernels.py attn_bwd > /dev/null 2>&1)
python3 -c "
import csv
print('$v', ' '.join('%s %.2f' % (r['Name'][32:38], float(r['AverageNs'])/1e6) for r in csv.DictReader(open('/tmp/p_x/stats_kernel_stats.csv')) if 'attn_bwd_d' in r['Name']))
"
done
