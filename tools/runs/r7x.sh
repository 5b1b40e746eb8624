#!/bin/bash
# round 5, call 7x: __graft_entry__.smoke() on the final build
O=gpurun_out/r7x; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "rc=$?"; tail -3 $O/smoke.txt
