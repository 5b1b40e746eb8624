set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3x
mkdir -p $O
for i in 1 2; do
python -m pytest tests -x -q -m gpu > $O/tests$i.log 2>&1 || { tail -40 $O/tests$i.log; exit 1; }
tail -1 $O/tests$i.log
done
for i in 1 2 3 4; do python tools/graph_vs_eager.py 2>&1 | grep -v amdgpu | head -1; done
