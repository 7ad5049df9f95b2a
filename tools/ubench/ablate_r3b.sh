#!/bin/bash
cd "$(dirname "$0")"
P="-DCONVB_NO_PRODUCER -DCONVB_NO_EPILOGUE"
V="$P;$P -DCONVB_NO_BARRIER;$P -DCONVB_NO_LDSREAD -DCONVB_NO_SCHED;$P -DCONVB_NO_BARRIER -DCONVB_NO_LDSREAD -DCONVB_NO_SCHED;$P -DCONVB_NO_SCHED"
echo "== consumer loop only, 128->256 3x3 (2,2,5,2) th=10 ck=16 (ideal MFMA 8.0 us + 2.4 launch)"
VARIANTS="$V" ./run_convb_ablate.sh 128 256 10 1 16 4 2 2
echo "== same, KS2"
K="-DKSPLIT=2 -DCFG=2,2,5,2,3,0,2"
VARIANTS="$K $P;$K $P -DCONVB_NO_BARRIER;$K $P -DCONVB_NO_BARRIER -DCONVB_NO_LDSREAD -DCONVB_NO_SCHED" ./run_convb_ablate.sh 128 256 10 1 16 4 2 2
echo "== consumer loop only, 128->768 (4,1,4,4) th=16 ck=16 (ideal MFMA 2 rounds x 12.8)"
C="-DCFG=4,1,4,4,3,0,1"
VARIANTS="$C $P;$C $P -DCONVB_NO_BARRIER;$C $P -DCONVB_NO_BARRIER -DCONVB_NO_LDSREAD -DCONVB_NO_SCHED" ./run_convb_ablate.sh 128 768 16 1 16 4 4 1
