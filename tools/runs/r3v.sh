set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3v
mkdir -p $O
rm -f $O/queues_sweep.txt
for q in 5 6 3; do
  echo "GPU_MAX_HW_QUEUES=$q" >> $O/queues_sweep.txt
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python tools/pair_graph_probe.py --n 2 >> $O/queues_sweep.txt 2>&1 || { tail -40 $O/queues_sweep.txt; exit 1; }
done
grep -v amdgpu.ids $O/queues_sweep.txt
