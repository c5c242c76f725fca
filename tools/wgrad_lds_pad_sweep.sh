F="--no-mc --no-roofline --no-second-dtype --no-cpu-baseline --no-pool-reader --steps 20 --warmup 3"
for P in 0 18000 34000 60000; do echo "== DASS_WX3_GROUP_LDS_PAD=$P"; DASS_WX3_GROUP_LDS_PAD=$P python bench.py $F 2>&1 | grep -o '"value": [0-9.]*, "unit": "images/s", "n_gpus"'; done
for P in 18000 34000; do echo "== pad $P + chunk 8"; DASS_WGRAD_CHUNK=8 DASS_WX3_GROUP_LDS_PAD=$P python bench.py $F 2>&1 | grep -o '"value": [0-9.]*, "unit": "images/s", "n_gpus"'; done
