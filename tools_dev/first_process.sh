for i in 1 2; do
GGA_BENCH_STEP_TIMES=1 python bench.py --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline > /tmp/b.out 2> /tmp/b.err
grep "host ms between" /tmp/b.err
tail -1 /tmp/b.out | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('PP', d['ms_per_step'], 'SECOND', d['second_trunk']['ms_per_step'])"
echo ---
done
