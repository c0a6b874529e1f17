# GPU side: SNOWGPU_FEW = 0 (k_power1) / 1 / 2 / 3: parity suite under 2 and 3, then same-box bench lines
export R=$GRAFT_REPO_ROOT; cd $R
for v in 2 3; do echo "== parity suite, SNOWGPU_FEW=$v"; SNOWGPU_FEW=$v timeout 600 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py 2>&1 | tail -3; done
bash scripts/ab_bench.sh "" "SNOWGPU_FEW=0" "SNOWGPU_FEW=1" "SNOWGPU_FEW=2" "SNOWGPU_FEW=3"
bash scripts/ab_bench.sh "--workload C2far" "SNOWGPU_FEW=0" "SNOWGPU_FEW=2" "SNOWGPU_FEW=3"
bash scripts/ab_bench.sh "--workload C4 --frames 128" "SNOWGPU_FEW=0" "SNOWGPU_FEW=2" "SNOWGPU_FEW=3"
