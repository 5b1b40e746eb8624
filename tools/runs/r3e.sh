set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e
mkdir -p $O
python tools/ablate_times.py > $O/ablate_cfg2.txt 2>&1; cat $O/ablate_cfg2.txt
