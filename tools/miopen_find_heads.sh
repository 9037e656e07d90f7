#!/bin/bash
# One-off on a GPU box: MIOpen solver search for the convolutions of the Mask R-CNN head stand-ins (bench.py --config maskrcnn),
# recorded into gpurun_out/miopen_db/ on top of the shipped find-db; copy the .ufdb.txt back into
# panoswintransformerobjectdetection_amd/miopen_db/ afterwards.
set -u
mkdir -p gpurun_out/miopen_db
cp panoswintransformerobjectdetection_amd/miopen_db/* gpurun_out/miopen_db/
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db
export PSWIN_MIOPEN_FIND=1
python bench.py --config maskrcnn --steps 3 --warmup 2 > gpurun_out/mr_find.json 2> gpurun_out/mr_find.err
echo "find run done: $(tail -c 300 gpurun_out/mr_find.json)"
ls -la gpurun_out/miopen_db
wc -l gpurun_out/miopen_db/*.ufdb.txt
unset PSWIN_MIOPEN_FIND
python bench.py --config maskrcnn --steps 5 --warmup 2 > gpurun_out/mr_after.json 2> gpurun_out/mr_after.err
python -c "
import json
d=json.loads(open('gpurun_out/mr_after.json').read().strip().splitlines()[-1]); print('after', d['value'], d['ms_per_step'], d['step_breakdown_ms'])"
