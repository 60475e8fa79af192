python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'], d['roofline']['ms_xv'], d['roofline']['ms_xtu'], d['roofline']['frac'])"
python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['value'], d['ms_per_step'], d['roofline']['ms_xv'], d['roofline']['ms_xtu'], d['roofline']['frac'])"
python bench.py --workload c5shard --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c5shard', d['value'], d['ms_per_step'], d['roofline']['ms_xv'], d['roofline']['ms_xtu'], d['roofline']['frac'])"
