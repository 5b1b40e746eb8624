cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/replay_latency.py 2>&1 | tail -1
TCE_ENCFORK=0 python tools/replay_latency.py 2>&1 | tail -1
python tools/replay_latency.py 2>&1 | tail -1
TCE_ENCFORK=0 python tools/replay_latency.py 2>&1 | tail -1
python tools/graph_vs_eager.py 2>&1 | grep -v amdgpu | head -1
