for w in base prev base prev; do
  if [ $w = base ]; then unset SMALLVCM_AMD_LIB; else export SMALLVCM_AMD_LIB=$PWD/scratch/lib_$w.so; fi
  python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['value'], d['ms_per_step'], r['light_kernel_ms'], r['per_kernel']['k_camera_trace']['ms'], r['connect_kernels_ms'], r['merge_kernel_ms'])"
done
