O=$PWD/differentiable-mel-spectrogram_amd/build/libdmel_hip_old.so
for r in 1 2; do
for c in c2 c3 c5 esc_n4096; do
echo "old: $(DMEL_LIB=$O python3 tools/ktime.py $c train 200 2>&1 | tail -1)"
echo "new: $(python3 tools/ktime.py $c train 200 2>&1 | tail -1)"
done
done
echo "old: $(DMEL_LIB=$O python3 tools/time_reference_shapes.py 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print({k:v['step_us'] for k,v in d.items()})")"
echo "new: $(python3 tools/time_reference_shapes.py 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print({k:v['step_us'] for k,v in d.items()})")"
