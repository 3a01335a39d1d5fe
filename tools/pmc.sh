#!/bin/bash
# Hardware-counter passes over a short bench run (each pass its own run, within the per-block counter limits).
out=gpurun_out/${1:-pmc}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ARGS="bench.py --no-cpu --steps 1 --warmup 0 --spp 16"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/a -o a -- python3 $ARGS > $out/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum -d $out/b -o b -- python3 $ARGS > $out/b.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/c -o c -- python3 $ARGS > $out/c.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/d -o d -- python3 $ARGS > $out/d.log 2>&1 || exit 1
ls $out/*/*.db
