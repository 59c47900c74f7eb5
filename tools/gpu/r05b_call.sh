# round 5, config-4 again: per-STEP kernel sums and spans (3 timed steps under rocprofv3, steps split at adam_kernel), with the
# board's clocks / power sampled once a second beside it -- r05a: kernel durations sum to 6.77 s per step, the step takes 7.24 s
tag=${1:-r05b}
R=$GRAFT_REPO_ROOT
cd $R
(for i in $(seq 1 150); do date +%s.%N; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|hotspot)" ; sleep 1; done) > gpurun_out/${tag}_smi.log 2>&1 &
SMI=$!
python bench.py --hw 832 992 --batch 2 --unroll 16 --steps 4 --warmup 1 --no-bf16 --no-infer --no-cpu-baseline --no-variants > gpurun_out/${tag}_f32_c4_bench_line.json 2> gpurun_out/${tag}_c4.err
python -c "
import json; d=json.load(open('gpurun_out/${tag}_f32_c4_bench_line.json')); print('c4 4 steps', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d.get('allocator'))"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_prof_c4 -- python $R/bench.py --hw 832 992 --batch 2 --unroll 16 --steps 3 --warmup 1 --no-bf16 --no-infer --no-cpu-baseline --no-variants > $R/gpurun_out/${tag}_c4_prof_line.json 2> $R/gpurun_out/${tag}_c4_prof.err)
python -c "
import json; d=json.load(open('gpurun_out/${tag}_c4_prof_line.json')); print('c4 under rocprof', d['ms_per_step'])"
kill $SMI 2>/dev/null
python tools/trace_gaps.py gpurun_out/${tag}_prof_c4 --step-marker adam_kernel --out gpurun_out/${tag}_c4_gaps.json | tee gpurun_out/${tag}_c4_gaps.txt
rm -rf gpurun_out/${tag}_prof_c4
grep -c sclk gpurun_out/${tag}_smi.log; grep -E "sclk|Power" gpurun_out/${tag}_smi.log | awk 'NR%6==1 || NR%6==2' | head -40
python tests/diag/diag3_gpu.py c1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_diag3_c1.log | head -70
timeout 2400 python -m pytest tests -q -x -s -m gpu > gpurun_out/${tag}_gpu_tests.log 2>&1; tail -4 gpurun_out/${tag}_gpu_tests.log
grep -E "train_step_parity|torch-fp32 vs|HIP fp32|worst max" gpurun_out/${tag}_gpu_tests.log | head -40
