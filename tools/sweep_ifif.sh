cd $GRAFT_REPO_ROOT
for cfg in "40 1 1" "40 32 24" "48 32 24" "56 32 24" "48 40 32" "56 24 16" "32 32 24" "48 32 1" "48 1 24"; do
  set -- $cfg
  echo "== refill $1 nodeMin $2 triMin $3"
  OKRT_FAST_REFILL=$1 OKRT_FAST_NODE_MIN=$2 OKRT_FAST_TRI_MIN=$3 bash tools/bench_all.sh sponza dragon80k dragon | awk '{printf "%s %s  ", $1, $2} END {print ""}' || exit 1
done
