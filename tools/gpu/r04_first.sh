# round 4, first call: the new full-frame oracle tests, the TensorFlow probe, the default bench line with its `variants` block
tag=${1:-r04a}
mkdir -p gpurun_out
python tools/tf_pin.py --out gpurun_out/tf_pin > gpurun_out/${tag}_tf_probe.log 2>&1; echo "tf_pin rc $?" >> gpurun_out/${tag}_tf_probe.log
python -c "import importlib.util as u; print([m for m in ('tensorflow','keras','cv2','tflite_runtime','jax') if u.find_spec(m)])" >> gpurun_out/${tag}_tf_probe.log 2>&1
cat gpurun_out/${tag}_tf_probe.log
( time python -m pytest tests/test_fullsize_gpu.py -q -m gpu -s -k "frame_size or config4_frame or every_gradient" ) > gpurun_out/${tag}_fullsize.log 2>&1
grep -v "^$" gpurun_out/${tag}_fullsize.log | cut -c1-400 | tail -60
python bench.py --steps 4 --warmup 2 > gpurun_out/${tag}_f32.json 2> gpurun_out/${tag}_f32.err; tail -2 gpurun_out/${tag}_f32.err
python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_f32.json'))
print('fp32', d['value'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d.get('inference'), (d.get('bf16_mode') or {}))
for k, v in (d.get('variants') or {}).items():
    for p in ('fp32', 'bf16'):
        r = v[p]
        print(k, p, r['ms_per_step'], r['step_tflops_achieved'], r['frac_of_peak'])
        for row in r['mfma_kernels'][:8]:
            print('    ', row)
print(d['cpu_baseline'])
PY
