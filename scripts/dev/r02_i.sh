R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/${1:-r02i}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_parity_large.py tests/test_gpu_backward.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-200
bash scripts/dev/variants.sh "--config cfg5a --steps 12 --warmup 2" "split=-DX1" 2>&1 | tail -2
bash scripts/dev/variants.sh "--config cfg5b --steps 4 --warmup 1" "split=-DX1" 2>&1 | tail -2
