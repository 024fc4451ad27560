#!/bin/bash
# wave-FFT dL/dx: parity subset, then timings (new path / LDS radix-2 path)
mkdir -p gpurun_out/r03j
timeout 900 python -m pytest tests -m gpu -x -q -k "xgrad or optional_gradients or unusual_shapes or dspec or full_window or random" 2>&1 | tail -15
python tools/time_backward_extras.py 2>&1 | tail -1 | tee gpurun_out/r03j/extras_wave.json
DMEL_XGRAD_LDS=1 python tools/time_backward_extras.py 2>&1 | tail -1 | tee gpurun_out/r03j/extras_lds.json
