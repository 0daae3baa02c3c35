for sk in 0 4160 0 4160 2112 16448; do
SMAC_ROW_SKEW=$sk timeout -k 10 200 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('skew $sk', round(d['value'],1), {a:k[a] for a in ('p2g','g2p','g2p_grad','p2g_grad','grid_op','reduce_agvout','sort')})"
done
