set -u
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
for rep in 1 2; do
  echo "== base-$rep"; $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print(d['value'], d['ms_per_step'], r['k_camera_trace']['ms'], r['k_connect_di+vc']['ms'], r['k_merge']['ms'])"
  echo "== prio-$rep"; SMALLVCM_AMD_STREAM_PRIO=1 $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print(d['value'], d['ms_per_step'], r['k_camera_trace']['ms'], r['k_connect_di+vc']['ms'], r['k_merge']['ms'])"
done
echo "== base-512"; $B --res 512 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
echo "== prio-512"; SMALLVCM_AMD_STREAM_PRIO=1 $B --res 512 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
