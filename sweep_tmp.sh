for s in 1 2 3 4 6; do
  timeout 200 python bench.py --cpu-frames 0 --streams $s > /tmp/o.txt 2>/tmp/e.txt || tail -5 /tmp/e.txt
  tail -1 /tmp/o.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('streams', c['streams'], round(d['value'],1), round(d['ms_per_step'],3))"
done
