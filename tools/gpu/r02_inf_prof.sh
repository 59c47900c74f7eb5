# streaming-inference kernel profile (B = 1, 256x256 frames): per-kernel / per-grid time, busy vs wall
tag=${1:-r02inf}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_$mode -- python $R/tools/inf_prof.py $mode > /dev/null 2>&1
  (cd $R && python tools/prof_summary.py gpurun_out/${tag}_prof_$mode gpurun_out/${tag}_${mode}_kernel_stats 80 | head -30
   python - <<PY
import glob, sqlite3
db = sorted(glob.glob('gpurun_out/${tag}_prof_$mode/**/*.db', recursive=True))[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select start, end from kernels order by start").fetchall()
n = len(rows); per = n // 24
last = rows[-8 * per:]
busy = sum(e - s for s, e in last); span = last[-1][1] - last[0][0]
print('$mode: %d launches/frame, last 8 frames: busy %.3f ms/frame, span %.3f ms/frame' % (per, busy / 8e6, span / 8e6))
PY
   rm -rf gpurun_out/${tag}_prof_$mode)
done
