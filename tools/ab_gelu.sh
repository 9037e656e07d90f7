run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$*', d['value'], 'pano/s', d['ms_per_step'], 'ms', {n:(k[n]['ms_per_step']) for n in ('pswin_gemm_nt','pswin_gemm_nt_gelu_bwd','pswin_bias_gelu_bwd','pswin_bias_gelu_fwd','lib_gemm_dgrad') if n in k})
"; }
run PSWIN_FUSED_GELU_BWD=1
run PSWIN_FUSED_GELU_BWD=0
run PSWIN_FUSED_GELU_BWD=1
run PSWIN_FUSED_GELU_BWD=0
