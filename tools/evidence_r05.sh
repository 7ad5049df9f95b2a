#!/bin/bash
# GPU box: the round's evidence logs -> gpurun_out/ev_r05/ (copied into profiles/r05_* afterwards)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev_r05; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gpu_tests.log
python -m pytest tests/test_gpu_headline_parity.py -q -s 2>&1 | grep -v "^$" | grep "frame\|worst\|per-frame\|passed\|failed\|golden\|held\|meet" | cut -c1-400 > $O/headline_parity.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/bench_torchrun_n1.json
python bench.py --stereo-only --no-cpu-baseline --steps 100 2>/dev/null > $O/bench_stereo_only.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > $O/bench_steps20.json   # the driver's K
{ for P in "--precision split --iters 1" "--precision bf16mix" "--precision fp16mix" "--precision bf16" "--precision fp16" "--precision split"; do echo "== bench.py --height 512 --width 640 $P"; python bench.py --height 512 --width 640 --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100 $P 2>/dev/null | cut -c1-260; done; echo "== bench.py --height 384 --width 1280"; python bench.py --height 384 --width 1280 --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100 2>/dev/null | cut -c1-260; for P in fp16mix split16 split; do echo "== bench.py (960x540) --precision $P"; python bench.py --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100 --precision $P 2>/dev/null | cut -c1-260; done; } > $O/cfg_variants.log
ls -la $O
