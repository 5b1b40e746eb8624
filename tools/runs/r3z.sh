cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in enc2 enc1 enc3 enc0 enc2; do echo LAT1_AT=$w; TCE_LAT1_AT=$w python tools/replay_latency.py --reps 80 2>&1 | tail -1 | cut -c1-60; done
