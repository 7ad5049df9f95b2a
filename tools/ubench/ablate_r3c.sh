#!/bin/bash
# epilogue store flavours: plain / nontemporal / sc1 (write-through) on the 128->256 3x3 layer
cd "$(dirname "$0")"
VARIANTS=";-DCONVB_NT_STORE;-DCONVB_SC1_STORE;-DCONVB_NO_EPILOGUE" ./run_convb_ablate.sh 128 256 10 1 16 4 2 2
VARIANTS="-DCFG=4,1,4,4,3,0,1;-DCFG=4,1,4,4,3,0,1 -DCONVB_NT_STORE;-DCFG=4,1,4,4,3,0,1 -DCONVB_SC1_STORE" ./run_convb_ablate.sh 128 768 16 1 16 4 4 1
