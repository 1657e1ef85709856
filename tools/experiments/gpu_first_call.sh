#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/firstcall; mkdir -p $O
sha256sum icde2019-gpu-join_amd/libhj.so > $O/out.txt
HJ_DEBUG=${HJ_DEBUG:-0} timeout 600 python3 tools/experiments/first_call.py ${SIZES:-27 31} 2>&1 | grep -v "parent " | tee -a $O/out.txt
