# same-box sweep of the attention work-item targets (items = chunks x bias windows x heads) inside the full training step
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$*', d['value'], 'pano/s', d['ms_per_step'], 'ms', {n:(k[n]['ms_per_step'], k[n]['GBps']) for n in ('pswin_attn_fwd','pswin_attn_bwd','pswin_attn_bwd_ex') if n in k})
"; }
run PSWIN_ATTN_TARGET_BWD=600
run PSWIN_ATTN_TARGET_BWD=1200
run PSWIN_ATTN_TARGET_BWD=2400
run PSWIN_ATTN_TARGET_BWD=4800
run PSWIN_ATTN_TARGET_BWD=600 PSWIN_ATTN_TARGET_FWD=2200
run PSWIN_ATTN_TARGET_BWD=600 PSWIN_ATTN_TARGET_FWD=4400
run PSWIN_ATTN_TARGET_BWD=600
