# One bench run (40 steps) per knob setting; prints ms/step. Usage: sweep_knobs.sh <tag> "ENV1=a ENV2=b" "ENV3=c" ...
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
cd $R
TAG=$1; shift
O=gpurun_out/${TAG}_sweep
mkdir -p $O
i=0
for kv in "default" "$@"; do
  i=$((i+1))
  if [ "$kv" = default ]; then envs=""; else envs="$kv"; fi
  env $envs timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-structured > $O/run$i.json 2> $O/run$i.err || { echo "[$kv] FAILED"; tail -3 $O/run$i.err; exit 1; }
  if grep -q "Memory access fault" $O/run$i.err; then echo "[$kv] faulted"; exit 1; fi
  python -c "import json; d=json.load(open('$O/run$i.json')); print('[$kv]', round(d['ms_per_step'],3), 'ms/step', round(d['value'],1), 'images/s; roofline frac', round(d['roofline']['frac'],4))"
done
