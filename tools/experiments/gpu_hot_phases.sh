#!/bin/bash
# per-phase times of the bypassing pass 1 (tools/experiments/hot_phases.py) with the -DHJ_STAMPS build in place of libhj.so
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/hotphases
P=icde2019-gpu-join_amd
cp $P/libhj.so $P/libhj_shipped.so
if [ "$SHIPPED" != 1 ]; then cp $P/libhj_stamps.so $P/libhj.so; fi   # SHIPPED=1: kernel times only (no phase stamps), the shipped registers
touch $P/libhj.so $P/bench
timeout 900 python tools/experiments/hot_phases.py 27 31 2>gpurun_out/hotphases/err.log | tee gpurun_out/hotphases/phases.jsonl
cp $P/libhj_shipped.so $P/libhj.so
tail -3 gpurun_out/hotphases/err.log
