set -u
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-configs --no-traffic --steps 12 --warmup 3"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print(d['value'], d['ms_per_step'], r['k_light_trace']['ms'], r['k_camera_trace']['ms'], r['k_connect_di+vc']['ms'], r['k_merge']['ms'])"; }
for rep in 1 2; do
  echo "== base-$rep"; $B 2>/dev/null | grep '^{' | p
  echo "== vc1-$rep"; SMALLVCM_AMD_VC_STREAM=1 $B 2>/dev/null | grep '^{' | p
  echo "== tb4608-$rep"; SMALLVCM_AMD_TASK_BLOCKS=4608 $B 2>/dev/null | grep '^{' | p
done
echo "== bpm"; $B --algo bpm 2>/dev/null | grep '^{' | p
echo "== s3"; $B --scene 3 --res 1024 2>/dev/null | grep '^{' | p
echo "== s3-vc0"; SMALLVCM_AMD_VC_STREAM=0 $B --scene 3 --res 1024 2>/dev/null | grep '^{' | p
echo "== 1024"; $B --res 1024 2>/dev/null | grep '^{' | p
echo "== 1024-vc0"; SMALLVCM_AMD_VC_STREAM=0 $B --res 1024 2>/dev/null | grep '^{' | p
