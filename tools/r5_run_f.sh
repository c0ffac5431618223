#!/bin/bash
# round-5 GPU call F: the fine-grid parity test; SQ counters of the FilterReg C4 kernels (is k_embed issue- or wait-bound?)
export TMPDIR=/tmp
out=gpurun_out/r5f
mkdir -p $out
timeout 300 python -m pytest tests/test_mfma_gpu.py -q --durations=5 > $out/pytest_mfma.log 2>&1
echo "mfma tests rc=$?" > $out/status.txt
tail -3 $out/pytest_mfma.log
c4="python bench.py --workload filterreg_500k --steps 20 --warmup 3"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $out/c4_sq -o b -- $c4 > $out/c4_sq.log 2>&1
python tools/rocpd_summary.py --pmc $(ls $out/c4_sq/*.db $out/c4_sq/*/*.db 2>/dev/null | head -1) > $out/r5_filterreg_500k_pmc_sq.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVES TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum --kernel-trace -d $out/c4_mem -o b -- $c4 > $out/c4_mem.log 2>&1
python tools/rocpd_summary.py --pmc $(ls $out/c4_mem/*.db $out/c4_mem/*/*.db 2>/dev/null | head -1) > $out/r5_filterreg_500k_pmc_mem.txt 2>&1
rm -rf $out/c4_sq $out/c4_mem
head -40 $out/r5_filterreg_500k_pmc_sq.txt
