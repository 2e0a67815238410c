# tools/dev/walk_bench.sh [variant ...]: the device walk on the bench's 1 M x 10 kb record stream, main library and variants
for v in main "$@"; do
  if [ $v = main ]; then unset DEXGPU_LIB; else export DEXGPU_LIB=tools/variants/libdexgpu_$v.so; fi
  timeout -k 5 300 python bench.py --no-cpu-baseline --only-main --steps 1 --warmup 1 --no-walk-index ${WB_ARGS} 2>gpurun_out/wb.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['device_walk']; print('$v', w.get('kernel_ms'), w.get('wall_ms'), w.get('pieces'), w.get('index_identical_to_the_encoders'), w.get('skipped'))"
done
