F="--no-mc --no-roofline --no-second-dtype --no-cpu-baseline --no-pool-reader --steps 20 --warmup 3"
python - <<'PY'
import torch
print("stream priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
PY
for P in 0 1 -1; do echo "== DASS_WGRAD_SIDE_PRIO=$P"; DASS_WGRAD_SIDE_PRIO=$P python bench.py $F 2>&1 | grep -o '"value": [0-9.]*, "unit": "images/s", "n_gpus"' ; done
echo "== side stream off"; DASS_WGRAD_SIDE=0 python bench.py $F 2>&1 | grep -o '"value": [0-9.]*, "unit": "images/s", "n_gpus"'
echo "== group tile 2 (3 stages, one wgrad workgroup per CU)"; DASS_WX3_GROUP_TILE=2 python bench.py $F 2>&1 | grep -o '"value": [0-9.]*, "unit": "images/s", "n_gpus"'
echo "== chunk 8"; DASS_WGRAD_CHUNK=8 python bench.py $F 2>&1 | grep -o '"value": [0-9.]*, "unit": "images/s", "n_gpus"'
echo "== chunk 32"; DASS_WGRAD_CHUNK=32 python bench.py $F 2>&1 | grep -o '"value": [0-9.]*, "unit": "images/s", "n_gpus"'
