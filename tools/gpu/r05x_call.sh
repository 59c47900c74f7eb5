# round 5: socket power and shader clock (rocm-smi, once a second) under the fp32 step, the bf16 step and the bf16x3 step -- the watts behind
# "power-limited" (DESIGN 3.3); each bench runs 40 steps in the background while the sampler runs
cd $GRAFT_REPO_ROOT
for prec in fp32 bf16x3 bf16; do
  steps=40; [ $prec = bf16 ] && steps=200
  python bench.py --precision $prec --steps $steps --warmup 3 --no-cpu-baseline --no-variants --no-infer --no-bf16 --no-x3 > gpurun_out/r05x_${prec}_line.json 2>/dev/null &
  pid=$!
  sleep 14
  for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showpower --showclocks --showtemp -d 0 2>/dev/null | grep -i "Package Power\|sclk\|junction" | sed "s/^/$prec /"
    sleep 1
  done
  wait $pid
  python -c "
import json; d=json.load(open('gpurun_out/r05x_${prec}_line.json')); print('$prec', d['ms_per_step'], d['value'])"
done 2>&1 | tee gpurun_out/r05x_power.log
