#!/bin/bash
# Memory-path diagnostics of the stage kernels on the bench's default workload, three counter passes in runs of their own:
# wave wait shares and vector-memory instruction counts (SQ), read latency and stalls at the L1 (TCP, TA), and what the
# L2 sends on (TCC). tools/prof_mem_diag.sh TAG -> gpurun_out/TAG/{sqmem,tcp,tcc}.txt
out=gpurun_out/${1:-diag}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="bench.py --no-cpu --steps 1 --warmup 0"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $out/sqmem -o a -- python3 $B > $out/sqmem.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -d $out/tcp -o a -- python3 $B > $out/tcp.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $out/tcc -o a -- python3 $B > $out/tcc.log 2>&1 || exit 1
for d in sqmem tcp tcc; do
  db=$(find $out/$d -name '*.db' | head -1)
  python3 profiles/pmc_summary.py $db > $out/$d.txt
  rm -rf $out/$d
done
