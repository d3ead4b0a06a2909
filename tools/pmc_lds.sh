#!/bin/bash
# SQ counter passes of the cfg2 bench (k_admm_lds): two rocprofv3 --pmc runs (8 SQ counters each), summary by make_pmc_summary.py
# usage: tools/pmc_lds.sh <out.txt> [lib]      (run from the repo root on the GPU box)
out=$1; lib=${2:-}
[ -n "$lib" ] && export MGADMM_LIB=$PWD/$lib
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/pmc1 /tmp/pmc2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d /tmp/pmc1 -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-cfg3-leg --no-prof > /tmp/pmc1.log 2>&1 || { tail -5 /tmp/pmc1.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT -d /tmp/pmc2 -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-cfg3-leg --no-prof > /tmp/pmc2.log 2>&1 || { tail -5 /tmp/pmc2.log; exit 1; }
python tools/make_pmc_summary.py $out /tmp/pmc1 /tmp/pmc2
