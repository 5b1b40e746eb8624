set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w
mkdir -p $O
rm -f $O/env_sweep.txt
run() {
  echo "== $*" >> $O/env_sweep.txt
  env "$@" timeout -k 10 200 python tools/replay_latency.py --reps 80 2>&1 | grep -v amdgpu.ids >> $O/env_sweep.txt || echo "FAILED rc=$?" >> $O/env_sweep.txt
}
run X=0
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=3
run DEBUG_HIP_FORCE_GRAPH_QUEUES=6
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_DYNAMIC_QUEUES=0
run GPU_STREAMOPS_CP_WAIT=0
run X=1
cat $O/env_sweep.txt
