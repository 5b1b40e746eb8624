cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo default; python tools/msda_bench.py 2>&1 | grep -v amdgpu
echo fewq-everywhere; TCE_MSDA_FEWQ_MAX=100000000 python tools/msda_bench.py 2>&1 | grep -v amdgpu
