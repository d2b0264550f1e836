ADSB_SCAN=sieve timeout -k 10 600 python -m pytest tests/ab_cases.py -x -q -m gpu -p no:cacheprovider > gpurun_out/sieve_cases.txt 2>&1; tail -15 gpurun_out/sieve_cases.txt; for s in sieve root sieve root; do ADSB_SCAN=$s timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-feed 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['cold_start']
print('$s', 'scan_ms', r['kernel_ms'], 'finish_ms', r['finish_order_ms'], 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'cold_scan_ms', c['kernel_ms'], 'frames', d['config']['frames_per_step'], 'ceil', r['read_ceiling_gbps'], d.get('parity_check'))" | tee -a gpurun_out/sieve_ab1.txt; done
