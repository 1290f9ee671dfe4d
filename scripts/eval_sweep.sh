# development: evaluation time (ms) of bench.py's eval leg for a few configs / kernel choices
ev() { python bench.py "$@" --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['eval']['ms'], d['eval']['ranks_per_s'])"; }
echo "C2 default $(ev)"
echo "C2 default $(ev)"
echo "C3 default $(ev --model DistMult --k 400 --eta 30 --dataset synth-wn18rr)"
echo "C4 default $(ev --dataset synth-yago310 --batch 8192)"
echo "HolE default $(ev --model HolE)"
echo "DistMult k200 default $(ev --model DistMult)"
