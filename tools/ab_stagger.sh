set -e
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d['final_loss'])"
OSI_STAGGER=1 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger', d['ms_per_step'], d['final_loss'])"
done
OSI_STAGGER=1 OSI_WGRAD_BLOCKS=4096 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger blocks4096', d['ms_per_step'], d['final_loss'])"
OSI_STAGGER=1 OSI_WGRAD_BLOCKS=4096 OSI_WGRAD3_BLOCKS=1536 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger blocks4096 w3 1536', d['ms_per_step'], d['final_loss'])"
OSI_STAGGER=1 OSI_SIDE_PRIO=n python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger normal prio', d['ms_per_step'], d['final_loss'])"
