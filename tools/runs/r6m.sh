#!/bin/bash
# round 5, call m: smoke() as the driver runs it, the driver's bench command, a replay soak on owned executables
O=gpurun_out/r6m; mkdir -p $O
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err; echo "bench rc=$?"; python -c "import json;d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['precision']['value_f32_exact'],d['graphs'])"
timeout -k 10 600 python tools/replay_soak.py > $O/replay_soak.txt 2>&1; echo "soak rc=$?"; tail -4 $O/replay_soak.txt
