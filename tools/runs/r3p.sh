set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p
mkdir -p $O
python tools/ablate_times.py > $O/ablate.txt 2>&1; cat $O/ablate.txt
echo "TEXT_LATE=0"; TCE_TEXT_LATE=0 python tools/replay_latency.py 2>&1 | tail -1
echo "TEXT_LATE=0 ablate swin2,swin3"; TCE_TEXT_LATE=0 TCE_ABLATE=swin2,swin3 python tools/replay_latency.py 2>&1 | tail -1
echo "text cached-like: ablate nothing, TEXT_LATE=1"; python tools/replay_latency.py 2>&1 | tail -1
