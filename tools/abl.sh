#!/bin/bash
# ablations of gmm_fused_kernel: AMX_FUSED_ABL bits 1 = no stores, 2 = no DMA after the first tile, 4 = no extras loop, 8 = no lockstep evaluation
for a in 0 1 2 4 8 15; do
  AMX_FUSED_ABL=$a python bench.py --workload gmm-train --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('abl', $a, d['stages']['gmm']['avg_ms'])"
done
