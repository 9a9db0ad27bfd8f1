for d in 5 6 8; do
python bench.py --no-cpu --no-pcie --depth $d --steps 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('depth', $d, d['value'], d['ms_per_step'])"
done
