cd $GRAFT_REPO_ROOT; for m in "" "--cached"; do timeout 120 python tools/r05/epoch_boundary_probe.py $m 2>&1 | tail -6; done
