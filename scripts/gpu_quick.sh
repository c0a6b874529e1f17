#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
grep -h "L6 case" gpurun_out/fullsize_parity.jsonl | tail -8 | cut -c1-330
