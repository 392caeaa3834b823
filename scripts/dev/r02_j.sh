R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/${1:-r02j}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_edge.py tests/test_gpu_parity_large.py tests/test_gpu_parity.py tests/test_gpu_backward.py tests/test_gpu_semantics.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-200
for cfg in cfg2 cfg3 cfg4 cfg5a; do
timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$cfg value %.0f ms/step %.4f' % (d['value'], d['ms_per_step']), {k: round(v,4) for k,v in d.get('phase_ms',{}).items() if k in ('sample_p3p','score','select_rescore','refine')}, [(k['stage'], round(k['avg_us'],1)) for k in d.get('kernels',[])])"
done
