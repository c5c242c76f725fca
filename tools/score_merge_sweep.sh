F="--no-cpu-baseline --no-roofline --no-second-dtype --no-pool-reader --steps 0 --warmup 0 --only mc"
for M in 1 2 3 4 6; do echo "== DASS_SCORE_MERGE=$M"; DASS_SCORE_MERGE=$M python bench.py $F --mc-batches 48 2>&1 | grep -o 'mc-dropout T=10: [0-9.]* pool images/s'; done
F2="--no-cpu-baseline --no-roofline --no-second-dtype --no-pool-reader --steps 0 --warmup 0 --only coreset"
for M in 1 2 3 4 6; do echo "== coreset DASS_SCORE_MERGE=$M"; DASS_SCORE_MERGE=$M python bench.py $F2 --mc-batches 48 2>&1 | grep -o 'core-set: features [0-9.]* pool images/s'; done
