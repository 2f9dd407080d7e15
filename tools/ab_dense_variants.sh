for v in base exp1 exp2 exp3 base; do
  if [ $v = base ]; then unset SR355_LIB_PATH; else export SR355_LIB_PATH=$PWD/super-resolution-images-for-3d-printing-defect-detection_amd/sr355/libsr355_$v.so; fi
  python bench.py --steps 2 --warmup 1 --no-rows --no-parity --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k={r['kernel']:r for r in d['kernels']}
print('$v', round(d['ms_per_step'],1), 'tail', k['dense_tail_fused<bf16,conv4+conv5>']['tflops'], 'pair', k['dense_pair_fused<bf16>']['tflops'], 'conv1', k['dense_conv1_stream<bf16,64->32>']['gbps'])"
done
