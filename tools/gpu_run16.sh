mkdir -p gpurun_out
python -m pytest tests/test_minigrid_view_gpu.py -m gpu -x -q 2>&1 | tail -2
for g in 1 2 4; do echo "G=$g"; TWG=$g python tools/view_bench.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d['kernel']=='mg_gen_obs': print(d['view'], d['see_through_walls'], round(d['ms']*1e3,1),'us', round(d['views_per_s']/1e9,2),'G/s', round(d['frac_of_8TBs'],3))
"; done
