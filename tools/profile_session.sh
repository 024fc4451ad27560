#!/bin/bash
# One profiling session on the GPU box (run through gpurun from the repo root): the bench line, kernel trace + stats of the bench
# command, FETCH_SIZE / WRITE_SIZE / SQ counter passes (each in its own run: --pmc is never combined with another trace domain) for
# config 2 AND for the other shapes (configs 3, 5, 4-on-one-GPU, the reference's n_fft 4096 / 8192 shapes), the reducers (one rank,
# two and four ranks sharing this GPU, the two-rank run five times), the optional gradients and the trainable-filterbank step.
# Raw output under gpurun_out/prof_$TAG; tools/summarize_profile.py turns it into the files committed under profiles/.
#   usage: tools/profile_session.sh <tag> [parts]      parts: any of  bench trace pmc shapes reducers extras stamps  (default: all)
TAG=${1:-r06}
PARTS=${2:-"bench trace pmc shapes reducers extras stamps"}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
BENCH="python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs"
EAGER="python3 bench.py --steps 30 --warmup 5 --mode eager --no-cpu-baseline --no-other-configs"
if has bench; then
  python3 bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
fi
if has trace; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/kt.log 2>&1
fi
if has pmc; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $EAGER > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $EAGER > $OUT/write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq1 -- $EAGER > $OUT/sq1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -- $EAGER > $OUT/sq2.log 2>&1
  # the kernel's length in CYCLES next to the vector pipe's busy cycles, one pass (roofline.valu_busy: VERDICT r05 #5)
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/clk -- $EAGER > $OUT/clk.log 2>&1
fi
if has shapes; then
  # the other transform sizes: trains of forward launches (tools/ktime.py): kernel trace, then the same three counter passes per shape
  for cfg in c3 c5 c4 esc_n4096 esc_n8192; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$cfg -- python3 tools/ktime.py $cfg train 40 > $OUT/kt_$cfg.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$cfg -- python3 tools/ktime.py $cfg train 10 > $OUT/fetch_$cfg.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$cfg -- python3 tools/ktime.py $cfg train 10 > $OUT/write_$cfg.log 2>&1
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $OUT/sq_$cfg -- python3 tools/ktime.py $cfg train 10 > $OUT/sq_$cfg.log 2>&1
    rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq2_$cfg -- python3 tools/ktime.py $cfg train 10 > $OUT/sq2_$cfg.log 2>&1
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/clk_$cfg -- python3 tools/ktime.py $cfg train 10 > $OUT/clk_$cfg.log 2>&1
    python3 tools/ktime.py $cfg train 2>&1 | tail -1 >> $OUT/ktime.txt
  done
  python3 tools/ktime.py c2 train 2>&1 | tail -1 >> $OUT/ktime.txt
  python3 tools/ktime.py c2 infer 2>&1 | tail -1 >> $OUT/ktime.txt
  python3 tools/time_reference_shapes.py > $OUT/reference_shapes.json 2> $OUT/reference_shapes.err
  python3 tools/batch_sweep.py > $OUT/batch_sweep.json 2> $OUT/batch_sweep.err
fi
if has reducers; then
  A="--steps 200 --warmup 20 --no-cpu-baseline --no-other-configs"
  DMEL_BENCH_FORCE_DIST=1 python3 bench.py $A --reducer rccl 2> /dev/null | grep "^{" > $OUT/bench_1rank_rccl.json
  DMEL_BENCH_FORCE_DIST=1 python3 bench.py $A --reducer mailbox 2> /dev/null | grep "^{" > $OUT/bench_1rank_mailbox.json
  python3 bench.py $A 2> /dev/null | grep "^{" > $OUT/bench_1rank_plain.json
  for i in 1 2 3 4 5; do
    DMEL_BENCH_SHARE_GPU=1 python3 bench.py $A --gpus 2 --reducer mailbox 2> /dev/null | grep "^{" > $OUT/bench_2ranks_one_gpu_mailbox_$i.json
  done
  DMEL_BENCH_SHARE_GPU=1 python3 bench.py $A --gpus 4 --reducer mailbox 2> /dev/null | grep "^{" > $OUT/bench_4ranks_one_gpu_mailbox.json
  DMEL_BENCH_SHARE_GPU=1 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --gpus 8 --reducer mailbox 2> /dev/null | grep "^{" > $OUT/bench_8ranks_one_gpu_mailbox.json
fi
if has extras; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_f2 -- python3 tools/time_backward_extras.py > $OUT/kt_f2.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_lfb -- python3 tools/time_learnable_fb.py > $OUT/kt_lfb.log 2>&1
  python3 tools/time_learnable_fb.py 2> /dev/null | grep "^{" > $OUT/learnable_fb.json
  python3 tools/time_fbgrad.py c2 c3 2> /dev/null | grep "^{" > $OUT/fbgrad.json
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq_lfb -- python3 tools/time_learnable_fb.py > $OUT/sq_lfb.log 2>&1
fi
if has stamps; then
  # phase stamps of the n_fft 1024 training kernel (diagnostic build made by `python tools/stamps.py build -DDMEL_ONLY_NFFT=1024` before the call)
  python3 tools/stamps.py run c2 > $OUT/stamps_c2.txt 2>&1
fi
ls $OUT | head -100
