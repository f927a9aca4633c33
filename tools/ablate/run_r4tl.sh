#!/bin/bash
# Phase timelines of the secondary configurations (diagnosis builds) -> gpurun_out/r4_tl_*.txt
set -e
mkdir -p gpurun_out
python -c "import __graft_entry__ as e; e.build()" > /dev/null
STAMPS_TABLE=cfg5 VND_WIN_STAMP_PHASES=1 timeout -k 10 300 python tools/win_stamps.py 16 960000 > gpurun_out/r4_tl_cfg5.txt 2>&1
STAMPS_TABLE=cfg3 VND_WIN_STAMP_PHASES=1 VND_SPEC_LA=1 timeout -k 10 300 python tools/win_stamps.py 24 2880000 > gpurun_out/r4_tl_cfg3.txt 2>&1
STAMPS_TABLE=cfg3 timeout -k 10 300 python tools/win_stamps.py 24 2880000 > gpurun_out/r4_tl_cfg3_lite.txt 2>&1
echo done
