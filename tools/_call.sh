export TMPDIR=/tmp
timeout 120 python tools/attn_bwd_check.py 2>&1 | grep "dq"
for sel in 3 2; do
(cd /tmp; rm -rf /tmp/p_$sel; TG_ATTN_BWD_DKDV=$sel timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$sel -o stats -- python $OLDPWD/tools/bench_kernels.py attn_bwd > /dev/null 2>&1)
python3 -c "
import csv
for r in csv.DictReader(open('/tmp/p_$sel/stats_kernel_stats.csv')):
    if 'attn_bwd' in r['Name']: print('sel $sel', r['Name'][23:50], float(r['AverageNs'])/1e6)
"
done
