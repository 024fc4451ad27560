#!/bin/bash
# One profiling session on the GPU box (run through gpurun from the repo root): kernel trace + stats of the bench command,
# FETCH_SIZE / WRITE_SIZE / SQ counter passes (each in its own run, --pmc never combined with other trace domains),
# the un-profiled bench line of the same build.  Raw output under gpurun_out/prof_$TAG; tools/summarize_profile.py turns
# it into the files committed under profiles/.
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs"
python3 bench.py --steps 300 --warmup 30 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 30 --warmup 5 --mode eager --no-cpu-baseline --no-other-configs > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 30 --warmup 5 --mode eager --no-cpu-baseline --no-other-configs > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq1 -- python3 bench.py --steps 30 --warmup 5 --mode eager --no-cpu-baseline --no-other-configs > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 30 --warmup 5 --mode eager --no-cpu-baseline --no-other-configs > $OUT/sq2.log 2>&1
# the n_fft 2048 kernel (configs 3 and 5): kernel trace of plain launch trains (registers, LDS, duration per dispatch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c3 -- python3 tools/ktime.py c3 train 60 > $OUT/kt_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c5 -- python3 tools/ktime.py c5 train 60 > $OUT/kt_c5.log 2>&1
# the shapes of the reference's own experiments (search_spaces.py): forward + dot per training step through the C ABI
python3 tools/time_reference_shapes.py > $OUT/reference_shapes.json 2> $OUT/reference_shapes.err
# the optional backward outputs (dL/dfb, dL/dx) and the global-memory FFT / chirp-z kernel: per-kernel durations
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_f2 -- python3 tools/time_backward_extras.py > $OUT/kt_f2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_big -- python3 tools/time_full_window.py > $OUT/kt_big.log 2>&1
# the step with each reducer on one rank (the exchange's own cost: one-rank communicator / self-addressed mailbox), and two ranks
# sharing this GPU through the mailbox (HIP IPC; RCCL refuses two ranks on one device)
A="--steps 200 --warmup 20 --no-cpu-baseline --no-other-configs"
DMEL_BENCH_FORCE_DIST=1 python3 bench.py $A --reducer rccl 2> /dev/null | grep "^{" > $OUT/bench_1rank_rccl.json
DMEL_BENCH_FORCE_DIST=1 python3 bench.py $A --reducer mailbox 2> /dev/null | grep "^{" > $OUT/bench_1rank_mailbox.json
DMEL_BENCH_SHARE_GPU=1 python3 bench.py $A --gpus 2 --reducer mailbox 2> /dev/null | grep "^{" > $OUT/bench_2ranks_one_gpu_mailbox.json
# phase stamps of the n_fft 1024 training kernel (diagnostic build made by `python tools/stamps.py build -DDMEL_ONLY_NFFT=1024` before the call)
python3 tools/stamps.py run c2 > $OUT/stamps_c2.txt 2>&1
ls -R $OUT | head -80
