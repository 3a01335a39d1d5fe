for args in "--lanes 1" "--lanes 1 --tune 8,24,4" "--lanes 1 --tune 8,24,5" "--lanes 2 --tune 8,24,4" "--lanes 2 --tune 8,24,2" "--lanes 2 --tune 4,24,3" "--lanes 2 --tune 16,24,3" "--lanes 2 --tune 8,16,3" "--lanes 2 --tune 8,32,3"; do
  timeout -k 10 120 python3 bench.py --no-cpu --steps 4 --warmup 1 $args > /tmp/o.json 2>/dev/null || exit 1
  python3 -c "
import json,sys;d=json.load(open('/tmp/o.json'));print('$args', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
