"""stdin: bench.py output; stdout: the figures an A/B run compares (wall per call, kernel time by in-run events, fraction, clock)."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        r = d.get('roofline') or {}
        print('ms_per_step %.4f  kernel_ms %s  frac %s  clock_mhz %s' % (d['ms_per_step'], r.get('avg_launch_ms'), r.get('frac'),
                                                                           (d.get('config') or {}).get('clock_mhz')))
