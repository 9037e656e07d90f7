# same-box A/B of the GEMM back ends inside the full training step (bench.py, 20 steps each)
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$*', d['value'], 'pano/s', d['ms_per_step'], 'ms', {n:(k[n]['ms_per_step'], k[n].get('TFLOPs')) for n in ('pswin_gemm_nt','pswin_gemm_tn','lib_gemm_fwd','lib_gemm_dgrad','lib_gemm_wgrad') if n in k})
"; }
run PSWIN_GEMM_NT=1 PSWIN_GEMM_TN=0
run PSWIN_GEMM_NT=0 PSWIN_GEMM_TN=0
run PSWIN_GEMM_NT=1 PSWIN_GEMM_TN=1
run PSWIN_GEMM_NT=1 PSWIN_GEMM_TN=0
run PSWIN_GEMM_NT=0 PSWIN_GEMM_TN=0
