cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo xcd bands; python tools/conv3_bench.py 2>&1 | grep -v amdgpu
echo plain; TCE_CONV3_XCD=0 python tools/conv3_bench.py 2>&1 | grep -v amdgpu
python tools/replay_latency.py 2>&1 | tail -1
TCE_CONV3_XCD=0 python tools/replay_latency.py 2>&1 | tail -1
