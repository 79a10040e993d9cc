cd smallvcm_amd/host
for rep in 1 2 3; do
for f in 1 2 3; do
  printf "inflight %d: " $f; ./vcm_render -s 1 -a vcm --res 2048 2048 --gpus 1 --shards 1 --inflight $f --collectives threads --same-window -i $((20*f)) --warmup 3 --json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['Mpaths_s'])"
done; done
