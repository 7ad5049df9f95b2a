#!/bin/bash
# GPU box: the round's evidence logs -> gpurun_out/ev_r04/ (copied into profiles/r04_* afterwards)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev_r04; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gpu_tests.log
python -m pytest tests/test_gpu_headline_parity.py -q -s 2>&1 | grep -v "^$" | grep "frame\|worst\|per-frame\|passed\|failed\|golden\|ill-cond" > $O/headline_parity.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/bench_torchrun_n1.json
python bench.py --stereo-only --no-cpu-baseline --steps 100 2>/dev/null > $O/bench_stereo_only.json
{ for P in "--precision split --iters 1" "--precision bf16mix" "--precision bf16" "--precision split"; do echo "== bench.py --height 512 --width 640 $P"; python bench.py --height 512 --width 640 --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100 $P 2>/dev/null | cut -c1-200; done; echo "== bench.py --height 384 --width 1280"; python bench.py --height 384 --width 1280 --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100 2>/dev/null | cut -c1-200; } > $O/cfg_variants.log
{ for v in 0 1; do for q in 96 128 192 256 512; do CODD_GN_PAIR=0 CODD_GN_MFMA=$v CODD_GN_Q4=$q python tools/time_gn.py 2>/dev/null; done; done; for q in 128 192 256; do CODD_GN_PAIR=1 CODD_GN_Q4=$q python tools/time_gn.py 2>/dev/null; done; tools/pmc_gn.sh; tools/ubench/gn_pmc.sh; } > $O/gn_builder.log 2>&1
{ cd tools/ubench; for rh in 12 18 36; do ./roll_ablate_CLK.bin 16 1 1 576 960 $rh; done; ./roll_ablate_CLK.bin 32 1 1 288 480 12; for v in base NOMFMA NOSTORE NOBARRIER NOLOAD; do echo == $v; ./roll_ablate_$v.bin 16 1 1 576 960 12; done; ./clk.bin; cd $R; } > $O/roll_ablation.log 2>&1
python tools/time_roll.py --bench > $O/roll_vs_tile.log 2>&1
ls -la $O
