export R=$GRAFT_REPO_ROOT; cd $R
BENCH_ARGS="--workload C3" bash scripts/_ab.sh 2 head hash
