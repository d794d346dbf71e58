set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_ragged -o t -- python3 tools/refops_probe.py 1 1 > gpurun_out/tr_ragged.log 2>&1
tail -1 gpurun_out/tr_ragged.log | cut -c1-100
python3 tools/busy_tail.py /tmp/tr_ragged 0.2
