set -e
cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
for f in 0 1; do
echo "== VLMC_RMS_NORM=$f"
VLMC_RMS_NORM=$f RANK_TIMELINE_ITERS=6 timeout -k 10 300 python tools/rank_timeline.py 1 2>&1 | grep prune_ms | tail -3 | cut -c1-200
done
done
