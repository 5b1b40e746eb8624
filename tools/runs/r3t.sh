cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t
mkdir -p $O
python bench.py --steps 60 --clips-per-graph 2 --no-cpu-baseline --no-roofline --no-variants > $O/g2.json 2> $O/g2.err
echo rc=$?; tail -5 $O/g2.err | cut -c1-200
python -c "import json;d=json.load(open('$O/g2.json'));print('G=2',d['value'],d['ms_per_step'])"
TCE_GROUP_BRANCHES=1 python bench.py --steps 60 --clips-per-graph 2 --no-cpu-baseline --no-roofline --no-variants > $O/g2b.json 2> $O/g2b.err
echo rc=$?; tail -3 $O/g2b.err | cut -c1-200
python -c "import json;d=json.load(open('$O/g2b.json'));print('G=2 branches',d['value'],d['ms_per_step'])"
