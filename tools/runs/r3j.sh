cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo few mix; python tools/concurrency_probe.py 2>&1 | grep -v amdgpu
echo few mix PROBE=1; TCE_FR_PROBE=1 python tools/concurrency_probe.py 2>&1 | grep -v amdgpu
echo tiled mix; PROBE_FEW=0 python tools/concurrency_probe.py 2>&1 | grep -v amdgpu
