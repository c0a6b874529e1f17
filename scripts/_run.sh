cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sampler.py -m gpu -q -x 2>&1 | tail -3
SERIAL=1 bash scripts/_ab.sh 2 s_win3 t_win2 t_win3 t_win4
