cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_prof; mkdir -p $O
(cd /tmp && timeout 120 rocprofv3 --list-avail > $O/avail.txt 2>&1 < /dev/null)
grep -i -o "UTCL[A-Za-z0-9_]*\|TCP_[A-Z0-9_]*MISS[A-Z0-9_]*\|TCC_[A-Z0-9_]*\(HIT\|MISS\)[A-Za-z0-9_]*\|TCP_TCC_READ[A-Z_]*\|TCP_PENDING[A-Z_]*\|TCP_TCC_[A-Z_]*LATENCY[A-Z_]*\|[A-Z_]*LATENCY[A-Z_]*" $O/avail.txt | sort -u | head -80
