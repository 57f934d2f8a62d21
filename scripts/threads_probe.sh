# How many host worker threads does a 16-CPU cgroup quota feed best?  (A worker spins; 16 spinning workers + the main
# thread + the runtime's helper threads exceed the quota and the whole group is throttled, cpu.stat tells.)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
lscpu | grep -i "model name\|^CPU(s)\|Thread(s) per core\|Socket\|NUMA node(s)" 
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "affinity: $(taskset -p $$ 2>/dev/null | cut -c1-120)"
stat() { grep "nr_throttled\|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo; }
for t in 16 15 14 12 10 8; do
  stat
  echo "== threads $t"
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs none --threads $t 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %.4g  ms/step %.1f' % (d['value'], d['ms_per_step']))"
done
stat
