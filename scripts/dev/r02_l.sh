R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/${1:-r02l}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_edge.py tests/test_gpu_parity.py tests/test_gpu_ref_golden.py tests/test_gpu_semantics.py tests/test_gpu_backward.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-200
bash scripts/dev/variants.sh "--config cfg2" "mapl=-DX1" "nomapl=-DESAC_NO_MAPL" 2>&1 | tail -4
