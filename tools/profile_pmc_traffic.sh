#!/bin/bash
# The two HBM-traffic counter passes of tools/profile_round.sh alone (FETCH_SIZE, WRITE_SIZE on eager steps) -> gpurun_out/prof/pmc_traffic.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
mkdir -p $O; rm -rf $O/fetch $O/write
PMCBENCH="python3 bench.py --no-capture --steps 2 --warmup 1 --no-cpu-baseline --kernel-steps 0 --sustain-steps 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $PMCBENCH > /dev/null 2> $O/fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $PMCBENCH > /dev/null 2> $O/write.err
echo "write pass done"
python3 tools/pmc_summary.py $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) $O/pmc_traffic.json > /dev/null
rm -rf $O/fetch $O/write
cat $O/pmc_traffic.json
