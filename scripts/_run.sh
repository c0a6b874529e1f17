export R=$GRAFT_REPO_ROOT; cd $R
bash scripts/_ab.sh 2 w3 w4 w2
