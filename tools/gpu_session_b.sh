#!/bin/bash
# round-3 session B: where the 32 x 32 plan at n_fft 1024 spends its time (phase stamps + SQ counters)
OUT=gpurun_out/r03b; mkdir -p $OUT
export TMPDIR=/tmp
B=$PWD/differentiable-mel-spectrogram_amd/build
timeout 300 python tools/stamps.py run c2 > $OUT/stamps_r32_c2.txt 2>&1
head -40 $OUT/stamps_r32_c2.txt
export DMEL_LIB=$B/libdmel_hip_r32.so
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq1 -- python3 tools/ktime.py c2 train 40 > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -- python3 tools/ktime.py c2 train 40 > $OUT/sq2.log 2>&1
python3 tools/pmc_sum.py $OUT/sq1 > $OUT/pmc_sq1.json; python3 tools/pmc_sum.py $OUT/sq2 > $OUT/pmc_sq2.json
cat $OUT/pmc_sq1.json $OUT/pmc_sq2.json
rm -rf $OUT/sq1 $OUT/sq2
