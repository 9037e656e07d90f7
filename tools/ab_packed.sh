python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "fused or window_attention" 2>&1 | tail -2
python tools/bench_fused.py 8 2>&1 | grep shift
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print(d['value'], 'pano/s', d['ms_per_step'], 'ms', {n:(k[n]['ms_per_step'], k[n]['avg_us']) for n in ('pswin_win_attn_fused_fwd','pswin_attn_bwd_ex','pswin_attn_fwd') if n in k}, d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['bound'])
"
