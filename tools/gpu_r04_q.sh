cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{ echo "== conv / detector / VAE parity"; timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_yolo_gpu.py tests/test_yolo11_gpu.py tests/test_rtdetr_gpu.py tests/test_flux_gpu.py tests/test_rcan_gpu.py -q -m gpu -p no:cacheprovider -k "conv or yolo or rtdetr or vae or rcan" 2>&1 | tail -4
  for rep in 1 2; do
    timeout 300 python bench.py --config 2 --steps 40 --warmup 6 --no-cpu-baseline --no-traffic 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('config 2', round(d['value'],3), 'pages/s', 'yolo net', round(c.get('detect_net_ms') or 0, 2), 'aux', c.get('detect_aux_ms'), 'sam enc', round(c['segment_ms']['encoder'],2))"
  done
  timeout 300 python bench.py --config 5 --stages inpaint --steps 3 --warmup 1 --no-cpu-baseline --no-traffic 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']['inpaint']; print('config 5 inpaint', round(d['ms_per_step'],1), 'ms  vae enc/dec', round(c['vae_encode_ms'],2), round(c['vae_decode_ms'],2))"
} > gpurun_out/r04_q.log 2>&1
cat gpurun_out/r04_q.log
