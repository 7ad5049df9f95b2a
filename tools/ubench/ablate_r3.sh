#!/bin/bash
# round 3: where does the time of the update-block convolutions go?  (run on the GPU box)
cd "$(dirname "$0")"
V=";-DCONVB_NO_PRODUCER;-DCONVB_NO_CONSUMER;-DCONVB_NO_EPILOGUE;-DCONVB_NO_PRODUCER -DCONVB_NO_EPILOGUE;-DCONVB_NO_PRODUCER -DCONVB_NO_CONSUMER -DCONVB_NO_EPILOGUE;-DCONVB_NO_SCHED"
echo "== 128->256 3x3, (2,2,5,2) th=10 ck=16"
VARIANTS="$V" ./run_convb_ablate.sh 128 256 10 1 16 4 2 2
echo "== 128->256 3x3, (2,2,5,2) th=10 ck=8"
VARIANTS=";-DCONVB_NO_PRODUCER" ./run_convb_ablate.sh 128 256 10 1 8 4 2 2
echo "== 256->256 3x3, (2,2,5,2) th=10 ck=16"
VARIANTS=";-DCONVB_NO_PRODUCER;-DCONVB_NO_EPILOGUE" ./run_convb_ablate.sh 256 256 10 1 16 4 2 2
echo "== 128->256 KS2"
VARIANTS="-DKSPLIT=2 -DCFG=2,2,5,2,3,0,2;-DKSPLIT=2 -DCFG=2,2,5,2,3,0,2 -DCONVB_NO_PRODUCER" ./run_convb_ablate.sh 128 256 10 1 16 4 2 2
echo "== 128->768 3x3 (4,1,4,4) th=16 ck=16"
VARIANTS="-DCFG=4,1,4,4,3,0,1;-DCFG=4,1,4,4,3,0,1 -DCONVB_NO_PRODUCER;-DCFG=4,1,4,4,3,0,1 -DCONVB_NO_EPILOGUE;-DCFG=4,1,4,4,3,0,1 -DCONVB_NO_PRODUCER -DCONVB_NO_EPILOGUE" ./run_convb_ablate.sh 128 768 16 1 16 4 4 1
