#!/bin/bash
# GPU box, LAST GPU job of the round (after the last kernel commit): the evidence logs of the FINAL tree -> gpurun_out/ev_r06/
# (copied into profiles/r06_* afterwards) + the rocprofv3 statistics and PMC passes (tools/profile_round.sh -> gpurun_out/prof_r06/).
# Fails (exit 3) when the kernel statistics do not contain the kernels the tree ships -- VERDICT r5: the round-5 summaries were
# captured three kernel commits before the final tree.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev_r06; mkdir -p $O; cd $R
git_head=$(cat $R/.evidence_head 2>/dev/null); echo "tree: ${git_head:-unknown}; library $(sha256sum codd_amd/csrc/libcodd_hip.so | cut -c1-16)" > $O/tree.txt
# the GPU suite in two halves (every test exactly once): the headline / recurrence / conditioned parity file with its per-frame log, then the rest
python -m pytest tests/test_gpu_headline_parity.py -m gpu -q -s > $O/headline_parity_full.log 2>&1
grep -v "^$" $O/headline_parity_full.log | grep "frame\|worst\|per-frame\|passed\|failed\|golden\|held\|meet\|oracle-vs-oracle" | cut -c1-400 > $O/headline_parity.log
{ tail -3 $O/headline_parity_full.log; python -m pytest tests -m gpu -q --ignore tests/test_gpu_headline_parity.py 2>&1 | tail -8; } > $O/gpu_tests.log
rm -f $O/headline_parity_full.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node 1 bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/bench_torchrun_n1.json
python bench.py --stereo-only --no-cpu-baseline --steps 100 2>/dev/null > $O/bench_stereo_only.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > $O/bench_steps20.json   # the driver's K
{ for P in "--precision split --iters 1" "--precision fp16mix" "--precision split"; do echo "== bench.py --height 512 --width 640 $P"; python bench.py --height 512 --width 640 --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100 $P 2>/dev/null | cut -c1-260; done; echo "== bench.py --height 384 --width 1280"; python bench.py --height 384 --width 1280 --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100 2>/dev/null | cut -c1-260; } > $O/cfg_variants.log
ROUND=r06 bash tools/profile_round.sh > $O/profile_round.log 2>&1
S=$R/gpurun_out/prof_r06/serial_kernel_stats.csv
rc=0
for K in se3_gn_build5_kernel cvx_upsample_se3w_kernel conv_bf16_kernel gn_heads_prep_kernel; do
  grep -q "$K" $S || { echo "EVIDENCE CHECK FAILED: $K not in $S" | tee -a $O/tree.txt; rc=3; }
done
for K in se3_gn_build3_kernel se3_gn_build_kernel hr_fuse_sum_kernel conv_chain_kernel; do
  grep -q "$K" $S && { echo "EVIDENCE CHECK FAILED: pruned kernel $K appears in $S" | tee -a $O/tree.txt; rc=3; }
done
echo "evidence check rc=$rc" >> $O/tree.txt
cat $O/tree.txt; tail -3 $O/gpu_tests.log; cut -c1-200 $O/bench_default.json; ls $O
exit $rc
