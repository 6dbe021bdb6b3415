for rep in 1 2; do for L in "$@"; do
  for p in 9815459 19000013; do MI355_ENGINE_LIB=$L python bench.py --exponent $p --no-cpu-baseline --steps 3000 --warmup 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], $p, d['ms_per_step'])"; done
  for odd in 9 3; do MI355_ENGINE_LIB=$L python bench.py --field crt --odd $odd --exponent 205271257 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], 'crt$odd', d['ms_per_step'])"; done
done; done
