for v in 0 1 0 1 0 1; do
HIP_FORCE_DEV_KERNARG=$v timeout -k 10 120 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('DEV_KERNARG=$v', round(d['value'],1), round(d['device_ms_per_step'],4), {a:k[a] for a in ('p2g','g2p','g2p_grad','p2g_grad','grid_op','contact','contact_grad','reduce_agvout','grid_op_grad','grid_checkpoint')})"
done
