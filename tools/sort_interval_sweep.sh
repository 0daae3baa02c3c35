for si in 1 4 8 16 32 64; do
timeout -k 10 300 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --sort-interval $si 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('interval $si', round(d['value'],1), 'dev_ms', round(d['device_ms_per_step'],4), {a:k.get(a) for a in ('p2g','g2p','g2p_grad','p2g_grad','sort')})"
done
