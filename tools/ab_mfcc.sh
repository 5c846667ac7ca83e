#!/bin/bash
# A/B of the MFCC kernel's occupancy on one box: --mfcc-tuning wgs=N caps the workgroups per CU (default: min(4, LDS limit))
for fe in mfcc mfplp plp; do
for wg in 2 3 4; do
    python bench.py --workload mfcc --front-end $fe --mfcc-tuning wgs=$wg --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$fe wgs<=$wg', d['stages']['mfcc']['avg_ms'])"
done; done
