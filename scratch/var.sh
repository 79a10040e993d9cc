for st in 8 32; do
  python bench.py --no-cpu-baseline --steps $st --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print($st, d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['iteration_ms'], {k:v['ms'] for k,v in r['per_kernel'].items()})"
done
