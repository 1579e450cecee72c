for i in 1 2 3; do for v in 0 1 2; do
  VMM_C3_PERSISTENT=$v python bench.py --no-train --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); f=d['denoiser_ms_by_kernel_family']; print('pw=$v', d['ms_per_step'], 'conv3x3', f['vmm_conv3x3_bf16x3'])"
done; done
