O=$PWD/differentiable-mel-spectrogram_amd/build/libdmel_hip_old.so
python -m pytest tests -x -q -m gpu > gpurun_out/gputests_full.txt 2>&1; grep -E "passed|failed" gpurun_out/gputests_full.txt | tail -2
for r in 1 2; do
echo "old: $(DMEL_LIB=$O python3 tools/time_reference_shapes.py 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print({k:v['step_us'] for k,v in d.items()})")"
echo "new: $(python3 tools/time_reference_shapes.py 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print({k:v['step_us'] for k,v in d.items()})")"
done
