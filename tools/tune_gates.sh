#!/bin/bash
# dev: tune the launch configurations missing from the shipped db at the four benchmarked shapes -> gpurun_out/tg_*.json
O=gpurun_out; rm -f $O/tg_*.json
timeout 300 python bench.py --steps 30 --no-cpu-baseline --fp32-steps 0 --tune-db $O/tg_a.json 2>/dev/null | tail -1 | cut -c1-120
timeout 300 python bench.py --steps 30 --no-cpu-baseline --fp32-steps 0 --height 384 --width 1280 --tune-db $O/tg_b.json 2>/dev/null | tail -1 | cut -c1-120
timeout 300 python bench.py --steps 30 --no-cpu-baseline --fp32-steps 0 --height 512 --width 640 --tune-db $O/tg_c.json 2>/dev/null | tail -1 | cut -c1-120
timeout 300 python bench.py --steps 30 --no-cpu-baseline --fp32-steps 0 --precision bf16 --height 512 --width 640 --tune-db $O/tg_d.json 2>/dev/null | tail -1 | cut -c1-120
