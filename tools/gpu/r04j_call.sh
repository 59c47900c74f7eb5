cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python tools/w_resident.py new 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04j_wres.log
