#!/bin/bash
# dev: frame rate under HIP runtime environment knobs (one bench.py run each)
OUT=gpurun_out/env_sweep.log; : > $OUT
run() { env "$@" python bench.py --steps 100 --no-cpu-baseline --fp32-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> $OUT 2>&1 || echo "$* FAILED" >> $OUT; }
run A=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run ROC_SYSTEM_SCOPE_SIGNAL=0
run DEBUG_HIP_DYNAMIC_QUEUES=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1000
run A=1
cat $OUT
